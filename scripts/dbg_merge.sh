R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/pcrcg_amd/libpcrcg_hip.so /tmp/cur.so
for name in notrack oldsplit cur; do
  if [ "$name" = "cur" ]; then cp /tmp/cur.so $R/pcrcg_amd/libpcrcg_hip.so; else cp $R/ab/$name.so $R/pcrcg_amd/libpcrcg_hip.so; fi
  echo "== $name"; python -m pytest tests/test_model_gpu.py -x -q -k "mini or kpfcnn or gcn or full" 2>&1 | grep -E "passed|failed|Error|assert " | tail -4
done
cp /tmp/cur.so $R/pcrcg_amd/libpcrcg_hip.so
