"""a17 pinned by the reference's OWN lines (VERDICT r1 missing #2 / weak #1c): tests/golden/ref_shim_shuffle.npz holds what
model/dataloader.py:121-125 (`video_src_ids`), :215-222 (`shuffle_chunks` re-ordering) and :224-257 (`shuffled_idx_img`)
produced when executed unmodified under the TF shim (tests/golden/make_shuffle_golden.py), with the random draws they
consumed.  Checked here, bit for bit: the oracle restatements, the product's host-side collate code, and (in
tests/test_index_gpu.py) the `merlot_shuffled_idx` kernel.  When /root/reference is present the generator is re-run live."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import index_oracle as ix
from oracle import input_oracle as io_

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(G, 'ref_shim_shuffle.npz'))


def cases(fx):
    for i in range(int(fx['shuffle/count'])):
        pre = f'shuffle/{i}/'
        get = lambda k: fx[pre + k] if (pre + k) in fx.files else None      # noqa: E731
        yield int(fx[pre + 'n']), int(fx[pre + 'B']), float(fx[pre + 'p']), get('num_shuffle'), get('u_select'), get('u_perm'), fx[pre + 'out']


def test_oracle_shuffled_idx_matches_the_reference_run(fx):
    seen_shuffled = 0
    for n, B, p, ns, us, up, out in cases(fx):
        assert np.array_equal(ix.shuffled_idx_img(B, n, p, ns, us, up), out)
        if ns is not None:
            assert np.array_equal(io_.shuffled_idx_img(ns, us, up, n), out)
            seen_shuffled += int((out >= 16).sum())
            assert out.max() < 16 + n and ((out < n) | (out >= 16)).all()       # shuffle_offset = 16 (:226)
            # a group with k drawn gets exactly k out-of-place markers
            assert np.array_equal((out.reshape(B, n) >= 16).sum(1), ns)
    assert seen_shuffled > 20


def test_oracle_video_src_ids_and_chunk_reordering_match_the_reference_run(fx):
    for rec, eoc, vs in zip(fx['vsrc/is_eoc_records'], fx['vsrc/is_eoc'], fx['vsrc/video_src_ids']):
        assert eoc[-1] == 1 and np.array_equal(eoc[:-1], rec[:-1])         # "Last segment is always end" (:122)
        assert np.array_equal(ix.video_src_ids(eoc), vs)
        chunks = [{'tokenized_cleaned_asr': [5], 'tokenized_raw_asr': [5], 'is_eoc': int(e)} for e in rec]
        assert np.array_equal(io_.text_features(chunks, True, len(rec), 32)[2], vs)
    idx = io_.shuffle_chunks_index(fx['chunks/video_src_ids'], fx['chunks/u'])
    assert np.array_equal(np.take_along_axis(fx['chunks/chunk_num_in'], idx, 1), fx['chunks/chunk_num_out'])
    assert np.array_equal(np.take_along_axis(fx['chunks/video_src_ids'], idx, 1), fx['chunks/video_src_ids_out'])
    # whole videos move together and keep their inner order
    for row_in, row_out in zip(fx['chunks/chunk_num_in'], fx['chunks/chunk_num_out']):
        assert sorted(row_in.tolist()) == sorted(row_out.tolist())


def test_product_host_code_matches_the_reference_run(fx):
    """merlot_amd.input_pipeline (the product's collate path) uses its own copies of the two index functions."""
    from merlot_amd import input_pipeline as ip
    idx = ip.shuffle_chunks_index(fx['chunks/video_src_ids'], fx['chunks/u'])
    assert np.array_equal(np.take_along_axis(fx['chunks/chunk_num_in'], idx, 1), fx['chunks/chunk_num_out'])


def test_reference_quirk_shuffle_without_chunk_shuffle_raises(fx):
    """model/dataloader.py:245 formats `k`, the loop variable of the shuffle_chunks branch (:219): with
    `shuffle_chunks: False` and image_shuffle_prob >= 1e-6 the reference's input_fn raises UnboundLocalError.  Recorded,
    not reproduced (merlot_amd.input_pipeline simply skips the chunk re-ordering)."""
    assert bool(fx['shuffle/unbound_k_without_shuffle_chunks'])


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference checkout not present (GPU box)')
def test_fixture_regenerates_identically_from_the_reference(tmp_path):
    env = dict(os.environ, MERLOT_GOLDEN_OUT=str(tmp_path))
    subprocess.check_call([sys.executable, os.path.join(G, 'make_shuffle_golden.py')], env=env, stdout=subprocess.DEVNULL)
    new, old = np.load(tmp_path / 'ref_shim_shuffle.npz'), np.load(os.path.join(G, 'ref_shim_shuffle.npz'))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        assert np.array_equal(new[k], old[k]), k
