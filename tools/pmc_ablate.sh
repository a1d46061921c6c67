#!/bin/bash
# GPU box helper: PMC counters of k_screen_encode per ablation config (tools/ablate.py order).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_ablate
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p0 -- python3 tools/ablate.py > $OUT/p0.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -- python3 tools/ablate.py > $OUT/p1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p0","p1"):
    rows=[]
    for path in glob.glob("gpurun_out/pmc_ablate/%s/**/*counter_collection.csv"%p, recursive=True):
        rows+=list(csv.DictReader(open(path)))
    enc=[r for r in rows if "k_screen_encode" in r["Kernel_Name"]]
    ids=sorted(set(int(r["Dispatch_Id"]) for r in enc))
    cfgs=["0x1","0x101","0x201","0x401","0x301","0x601","0x701"]
    per=len(ids)//len(cfgs)
    tab=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in enc:
        k=ids.index(int(r["Dispatch_Id"]))//per
        tab[cfgs[min(k,len(cfgs)-1)]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names=sorted({c for t in tab.values() for c in t})
    print("%-8s"%"cfg"+"".join("%22s"%n for n in names))
    for c in cfgs:
        print("%-8s"%c+"".join("%22.0f"%(sum(tab[c][n])/max(1,len(tab[c][n]))) for n in names))
PY
