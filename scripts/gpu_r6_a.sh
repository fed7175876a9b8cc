#!/bin/bash
# round 6, call 1: (1) the tests of the ADVICE r5 fixes; (2) VERDICT r5 #1(b) as its cheapest build -- two co-resident 4-wave workgroups per CU on
# 128 x 256 tiles (gemm_nt_ring_kernel<RingQ>, experiments id 23) against the ping-pong kernel (id 22), mirrored order, both batch sizes; (3) the TCC
# traffic file with the joint / text-only attention kernels and the LayerNorm kernels added (VERDICT r5 #3 "measure first")
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_persist_gpu.py tests/test_model_gpu.py tests/test_reference_shim_gpu.py -x -q -m gpu -k "persist or alpha or sort_story or story or column_sums or k_tiles" 2>&1 | grep -v Warning | tail -4 | tee gpurun_out/r06_a_tests.txt
for T in 101376 405504; do
  echo "== T=$T  CFGS=22,23,23,22 (22 = ping-pong 256x256 persistent; 23 = 128x256, 4 waves, two workgroups per CU)" | tee -a gpurun_out/r06_a_two_wg_ab.txt
  T=$T CFGS=22,23,23,22 timeout 900 python scripts/exp_skew.py 2>&1 | grep -v Warning | tee -a gpurun_out/r06_a_two_wg_ab.txt
done
bash scripts/gpu_traffic.sh > /dev/null 2>&1; grep "HBM_MB\|hash" gpurun_out/r06_traffic.txt
