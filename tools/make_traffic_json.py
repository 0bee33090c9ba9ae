"""profiles/traffic.json from two rocprofv3 counter passes of the bench command (tools/refresh_profiles.sh: pmc1 = FETCH_SIZE ..., pmc2 =
WRITE_SIZE; `--proteins 2048` = 1 048 576 rows = whole chunks of bench.py's default size, MDF_DEFAULT_CHUNK_ROWS), stamped with the library's mdf_version() -- a hash of the GraphConv
kernels' source.  bench.py reports these bytes as `roofline*.traffic` only while the stamp matches the library it runs.
    python3 tools/make_traffic_json.py gpurun_out/refresh/pmc1 gpurun_out/refresh/pmc2 ROUND > profiles/traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))


def means(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    tot, n = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mdf::", "")
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(n[k]) for k in tot}


def pick(m, prefix):
    hit = [k for k in m if k.startswith(prefix)]
    assert len(hit) == 1, (prefix, sorted(m))
    return m[hit[0]]


def main():
    from mDeepFRI import _hip
    fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
    rows = _hip.default_chunk_rows()   # (the passes run bench.py with its default --chunk-rows, which is this number)
    out = {"_comment": f"HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; {rows} residue rows per launch, configs[2] "
                       "inputs). bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 rocprofv3 tallies "
                       "128-B read requests at 64 B); WRITE_SIZE reproduces the algorithmic store bytes (131072 KB for a 128 MiB slab) and is used as is. "
                       "`library` is mdf_version() of the build the passes ran on: bench.py drops `traffic` for any other build.",
           "round": int(sys.argv[3]), "library": _hip.lib().mdf_version().decode(), "rows_per_launch": rows}
    # the A.X launches of the headline workload (L = 512) all run the matrix-pipe kernel k_aggregate_mfma<2>; a build or workload that still
    # runs the CSR gather there is picked up under the same key
    # (k_aggregate_mfma<2, false> is the A.X kernel proper; <2, true> also makes layer 1 and is listed beside it)
    agg_name = next((k for k in fetch if k.startswith("k_aggregate_mfma<2") and "true" not in k and "(bool)1" not in k), None) or next(k for k in fetch if k.startswith("k_aggregate<512>"))
    fused_name = next((k for k in fetch if k.startswith("k_aggregate_mfma<2") and ("true" in k or "(bool)1" in k)), None)
    # the H.W launches: k_gemm_bf16x6<EPI> (default) or k_gemm_f32<EPI> (MDFRI_HW_PIPE=f32); EPI 0 stores the layer output, 1 only pools
    gemm = "k_gemm_bf16x6" if any(k.startswith("k_gemm_bf16x6<") for k in fetch) else "k_gemm_f32"
    for name, prefix in (("k_aggregate", agg_name), (f"{gemm}<0>", f"{gemm}<(Epilogue)0"), (f"{gemm}<1>", f"{gemm}<(Epilogue)1")):
        try:
            f, w = pick(fetch, prefix), pick(write, prefix)
        except AssertionError:
            alt = prefix.replace("(Epilogue)", "")
            f, w = pick(fetch, alt), pick(write, alt)
        out[name] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1), "bytes": int((2 * f + w) * 1024)}
    out["k_aggregate"]["kernel"] = agg_name
    if fused_name:
        f, w = pick(fetch, fused_name), pick(write, fused_name)
        out["k_aggregate_with_layer1"] = {"kernel": fused_name, "fetch_kb": round(f, 1), "write_kb": round(w, 1), "bytes": int((2 * f + w) * 1024)}
    out["gemm_kernel"] = gemm
    out["gemm_mean_bytes"] = (out[f"{gemm}<0>"]["bytes"] + out[f"{gemm}<1>"]["bytes"]) // 2
    print(json.dumps(out, indent=2))


main()
