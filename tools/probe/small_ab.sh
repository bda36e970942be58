export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
for f in 1 2 4; do for m in default NO_WIN NO_FLOW_SEAM NO_MERGED_RB NO_LN_FUSE NO_WIDE_POST; do
  if [ $m = default ]; then v=X; else v=PIPER_HIP_$m; fi
  echo "$m f=$f: $(env $v=1 timeout -k 10 100 python tools/profile_steps.py --factor $f 2>/dev/null | grep 'graph gpu_ms' | cut -c1-90)"
done; done
