export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/wc
run() { local name=$1; shift; local args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout -k 10 150 python tools/profile_steps.py "${args[@]}" > gpurun_out/wc/$name.txt 2>&1
  grep -q "Memory access fault" gpurun_out/wc/$name.txt && exit 1
  echo "$name: $(head -1 gpurun_out/wc/$name.txt)"; grep "convT\|dec.s0.rb" gpurun_out/wc/$name.txt | head -6
}
for cfg in default 4,1,1 2,2,1 1,4,1 2,1,2 1,2,2 1,1,4; do
  if [ $cfg = default ]; then run f8_default --factor 8 -- X=1; else run f8_$cfg --factor 8 -- PIPER_HIP_WIN_CFG=$cfg; fi
done
for cfg in default 2,2,1 1,4,1; do
  if [ $cfg = default ]; then run f64_default --factor 64 -- X=1; else run f64_$cfg --factor 64 -- PIPER_HIP_WIN_CFG=$cfg; fi
done
for g in 1 2.5 10; do run f8_pipemin$g --factor 8 -- PIPER_HIP_PIPE_MIN_GFLOP=$g; done
for g in 1 2.5 10; do run high_pipemin$g --quality high -- PIPER_HIP_PIPE_MIN_GFLOP=$g; done
