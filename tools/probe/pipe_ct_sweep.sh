export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/pct
run() { # name, args..., -- env...
  local name=$1; shift
  local args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout -k 10 150 python tools/profile_steps.py "${args[@]}" > gpurun_out/pct/$name.txt 2>&1
  grep -q "Memory access fault" gpurun_out/pct/$name.txt && exit 1
  echo "$name: $(head -1 gpurun_out/pct/$name.txt)"; grep "convT" gpurun_out/pct/$name.txt
}
run f64_pipe --factor 64 -- X=1
run f64_win --factor 64 -- PIPER_HIP_PIPE_CT_MIN_GFLOP=1000
run high_pipe --quality high -- X=1
run high_win --quality high -- PIPER_HIP_PIPE_CT_MIN_GFLOP=1000
run b8_pipe --batch 8 -- X=1
run b8_win --batch 8 -- PIPER_HIP_PIPE_CT_MIN_GFLOP=1000
run high64_pipe --quality high --factor 32 -- X=1
run high64_win --quality high --factor 32 -- PIPER_HIP_PIPE_CT_MIN_GFLOP=1000
