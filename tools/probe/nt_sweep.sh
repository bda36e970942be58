export PIPER_HIP_TUNING=1  # the library honours PIPER_HIP_* switches only with this set
set -e
mkdir -p gpurun_out/nt
run() { # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 120 python tools/profile_steps.py --factor 64 > gpurun_out/nt/f64_$name.txt 2>&1
  grep -q "Memory access fault" gpurun_out/nt/f64_$name.txt && exit 1
  echo "$name: $(head -1 gpurun_out/nt/f64_$name.txt)"; grep "flow3.wn1\|enc1.ln1_ffn1\|enc1.ffn2\|enc1.ln2_qkv\|enc1.o_add" gpurun_out/nt/f64_$name.txt
}
run default X=1
run nt2 PIPER_HIP_NT=2 PIPER_HIP_NT_MIN_L=1024 PIPER_HIP_KS_ANY_NT=1
run nt2_noks PIPER_HIP_NT=2 PIPER_HIP_NT_MIN_L=1024
run nt4 PIPER_HIP_NT=4 PIPER_HIP_NT_MIN_L=1024 PIPER_HIP_KS_ANY_NT=1
