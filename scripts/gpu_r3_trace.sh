for ph in 0 1; do echo "== MERLOT_P8_PH2=$ph"; MERLOT_P8_PH2=$ph python scripts/exp_p8_trace.py 2>&1 | grep -v amdgpu.ids; done
