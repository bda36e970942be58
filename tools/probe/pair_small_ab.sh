export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
for f in 1 2 4 8; do for m in pair nopair; do
  if [ $m = nopair ]; then export PIPER_HIP_NO_RB_PAIR=1; else unset PIPER_HIP_NO_RB_PAIR; fi
  echo "$m f=$f: $(timeout -k 10 100 python tools/profile_steps.py --factor $f 2>/dev/null | grep 'graph gpu_ms' | cut -c1-90)"
done; done
