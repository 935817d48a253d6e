# Where the Track X convolution GEMMs stall (CIFAR shape, fp32): three PMC passes over bench_convnet.py, counters summed per kernel name.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/px_$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/px_$tag -- python3 $R/bench_convnet.py --config cifar --steps 10 --warmup 4 > /dev/null 2> $R/gpurun_out/px_$tag.err || { tail -5 $R/gpurun_out/px_$tag.err; continue; }
  python3 - $R/gpurun_out/px_$tag <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    if "conv" in k: print(k, {c: round(v / max(1, cnt[(k, c)])) for c, v in acc[k].items()})
PY
  find $R/gpurun_out/px_$tag -type f -delete
done
