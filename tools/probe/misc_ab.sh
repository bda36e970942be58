export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
run() { local name=$1; shift; local args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  echo "$name [${args[*]}]: $(env "$@" timeout -k 10 120 python tools/profile_steps.py "${args[@]}" 2>&1 | head -1)"; }
for rep in 1 2; do
  for cfg in "--factor 16" "--factor 24" "--factor 32" "--factor 8 --batch 2" "--factor 8 --batch 4"; do
    run default $cfg -- X=1
    run nolnfuse $cfg -- PIPER_HIP_NO_LN_FUSE=1
  done
done
