"""CPU oracle for the MERLOT pretraining hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``merlot_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / timed baseline.

PARITY STATUS.  The reference (rowanz/merlot) is TF-1.15 graph code, ships no
tests or golden vectors, and tensorflow (pinned ``tensorflow==1.15.5``,
requirements.txt:70) is neither installed nor installable here, so the
restatement cannot be checked against TensorFlow's kernels: for the semantics
of the individual ``tf.*`` primitives **parity is unpinned**.  Everything above
the primitives IS pinned, by outputs of the reference's own code run in the
build container:

* ``tests/golden/ref_shim_*.npz`` -- the UNMODIFIED reference modules
  (model/modeling.py, utils/transformer.py, utils/vision_transformer.py,
  utils/model_utils.py, utils/optimization.py, and the ``model_fn`` of
  downstream/sort_story/get_zero_shot_logits.py) imported from /root/reference
  and executed under ``oracle/tf_shim.py`` (an eager torch-CPU stand-in for the
  ~120 tf symbols they use; SURVEY.md 8c "optional stronger oracle").  That
  runs the reference's control flow, variable scoping/naming, every
  reshape/tile/concat order, mask_inputs, the three losses, the 2-replica
  cross_replica_sum path and the AdamOptimizer update as written; the
  restatement agrees with it to ~1e-7 (outputs), <=1e-5 (all 115 gradients),
  bit-exact (masked ids/idx, bf16 optimizer states).
  ``tests/test_reference_shim.py`` checks this from the fixtures everywhere and
  re-runs the reference live where /root/reference exists.
* ``tests/golden/ref_shim_input_pipeline.npz`` -- the input-pipeline functions
  of utils/model_utils.py (resize_and_pad, lightweight_image_augment,
  encode_string, pad_to_fixed_size, sample_bernoulli) executed the same way
  (``tests/golden/make_input_golden.py``); ``oracle/input_oracle.py`` agrees to
  <= 2e-7.  The four tf.image resize kernels underneath are restated from the
  TF-1.15 sources (**parity unpinned** for those, as for every primitive).
* ``tests/golden/sort_story_ref.npz`` / ``tokenizer_ref.npz`` -- pure-python
  reference functions (score_permutations.py, encoder.py constants) executed
  directly (``tests/golden/make_golden.py``).
"""
