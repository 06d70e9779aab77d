cd $GRAFT_REPO_ROOT
ALL=$(python -c "import bench; print(' '.join(sorted(bench.COMMITTED_SWITCHES['bf16'])))" 2>/dev/null)
for drop in NONE MDETR_WFOLD MDETR_RELU_PREMASK; do
  L=""; for k in $ALL; do [ "$k" != "$drop" ] && L="$L $k=1"; done
  echo "== without $drop"
  env $L timeout 400 python -m pytest tests/test_trainer_gpu.py -m gpu -x -q -p no:cacheprovider -k "replayed_training_iteration" 2>&1 | grep -E "passed|failed|graph \[" | cut -c1-300
done
