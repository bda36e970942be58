export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
for f in 16 24 32 40 64; do
  for mt in 100000 512 256 128; do
    echo "min_t=$mt f=$f: $(PIPER_HIP_ATT_SPLIT_MIN_T=$mt timeout -k 10 100 python tools/profile_steps.py --factor $f 2>/dev/null | grep 'graph gpu_ms\|enc1.rel_att' | tr '\n' ' ' | cut -c1-150)"
  done
done
