#!/bin/bash
# Dynamic instruction counts per wave and kernel (PMC, own passes): what does a wave of each kernel actually issue?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_insts
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-scale-bench --no-profile --steps 20 --warmup 3"
i=0
for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/bench.py" $COMMON > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import sys,glob,csv,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r"void ph::detail::|void \(anonymous namespace\)::|ph::\(anonymous namespace\)::|\(ph::ConvArgs.*|\(.*","",r["Kernel_Name"])
        e=acc[n][r["Counter_Name"]]; e[0]+=float(r["Counter_Value"]); e[1]+=1
ctrs=sorted({c for k in acc for c in acc[k]})
print("%-52s %7s "%("kernel","launch")+" ".join("%12s"%c.replace("SQ_INSTS_","") for c in ctrs))
for k,v in sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_INSTS_VALU",[0,0])[0]):
    w=v.get("SQ_WAVES",[0,1]); nl=w[1]; waves=w[0]/max(nl,1)
    print("%-52s %7d "%(k[:52],nl)+" ".join("%12.1f"%(v[c][0]/max(v[c][1],1)/(waves if (c!="SQ_WAVES" and waves) else 1)) if c in v else "%12s"%"-" for c in ctrs))
PY
rm -rf "$OUT"  # the table above is what is kept
