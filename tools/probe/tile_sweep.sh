export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/tl
run() { local name=$1; shift; local args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout -k 10 150 python tools/profile_steps.py "${args[@]}" > gpurun_out/tl/$name.txt 2>&1
  grep -q "Memory access fault" gpurun_out/tl/$name.txt && exit 1
  echo "$name: $(head -1 gpurun_out/tl/$name.txt)"; grep "flow3.wn1\|flow3.pre\|flow0.post\|conv_pre" gpurun_out/tl/$name.txt | head -6
}
run f64_default --factor 64 -- X=1
run f64_tile1 --factor 64 -- PIPER_HIP_TILE=1
run f64_tile1_mt2 --factor 64 -- PIPER_HIP_TILE=1 PIPER_HIP_TILE_MT=2
run f64_tile1_mt2_ntw1 --factor 64 -- PIPER_HIP_TILE=1 PIPER_HIP_TILE_MT=2 PIPER_HIP_TILE_NTW=1
run f64_tile1_mt1 --factor 64 -- PIPER_HIP_TILE=1 PIPER_HIP_TILE_MT=1 PIPER_HIP_TILE_NTW=1
