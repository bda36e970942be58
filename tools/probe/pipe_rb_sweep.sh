export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/prb
run() { local name=$1; shift; local args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout -k 10 150 python tools/profile_steps.py "${args[@]}" > gpurun_out/prb/$name.txt 2>&1
  grep -q "Memory access fault" gpurun_out/prb/$name.txt && exit 1
  echo "$name: $(head -1 gpurun_out/prb/$name.txt)"; grep "dec.s0.rb\|dec.s1.rb" gpurun_out/prb/$name.txt | head -6
}
run high_rbpipe --quality high -- PIPER_HIP_PIPE_CT_MIN_GFLOP=1000
run high_nopipe --quality high -- PIPER_HIP_NO_PIPE=1
run f64_rbpipe --factor 64 -- PIPER_HIP_PIPE_CT_MIN_GFLOP=1000
run f64_nopipe --factor 64 -- PIPER_HIP_NO_PIPE=1
run b8_rbpipe --batch 8 -- PIPER_HIP_PIPE_CT_MIN_GFLOP=1000
run b8_nopipe --batch 8 -- PIPER_HIP_NO_PIPE=1
