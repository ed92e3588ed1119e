for cfg in "" "AVD_WIDE_FUSED_DX=0" "AVD_WIDE_FUSED_DELTA=0" "AVD_WIDE_FUSED_DW=0"; do
  f=0
  for i in 1 2 3 4 5 6 7 8; do
    env $cfg python -m pytest tests/test_gpu_configs_full.py -x -q -m gpu -k "full_size_mean" 2>&1 | grep -q "1 passed" || f=$((f+1))
  done
  echo "cfg=[$cfg] failures=$f/8"
done
