"""PMC passes of one kernel (GPU box): python tools/pmc.py <kernel-name-substring> <out.json> -- <command ...>
One rocprofv3 run per counter set (--kernel-trace only next to --pmc, as the pool requires); averages over the matching dispatches."""
import collections, csv, glob, json, os, shutil, subprocess, sys

SETS = [
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU",
    "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD",
    "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH",
    "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES",
]


def main():
    match, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    res = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k, cs in enumerate(SETS):
        d = f"/tmp/pmc_{os.getpid()}_{k}"
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--pmc", *cs.split(), "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "out", "--"] + cmd,
                           cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        f = glob.glob(d + "/**/out_counter_collection.csv", recursive=True)
        if not f:
            res[f"set{k}_error"] = (r.stderr or r.stdout)[-400:]
            continue
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f[0])):
            if match in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, v in agg.items():
            res[name] = sum(v) / len(v)
            res["dispatches"] = len(v)
    print(json.dumps(res, indent=1))
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
