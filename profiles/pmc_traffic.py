"""Turn the two rocprofv3 PMC passes of `bench.py --pmc-workload` into profiles/rNN_traffic.json (4th argument, default profiles/r03_traffic.json).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --pmc-workload ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --pmc-workload ...
    python3 profiles/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv

Units and corrections follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950
FETCH_SIZE reports half the bytes of wide (16 B per lane) coalesced reads, which is what these kernels issue, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  The workload runs UNet + decode twice; only the second pass
(after the warm-up) is counted, split into the UNet step and the decode at the decode's first kernel
(conv3x3_in_kernel<4>).
"""
import csv, json, re, sys, collections


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0], float(r["Counter_Value"])) for r in rows]


def second_pass(seq):
    starts = [i for i, (k, _) in enumerate(seq) if k.startswith("timestep_embedding_kernel")]
    assert len(starts) >= 2, "expected two UNet passes"
    seq = seq[starts[-1]:]
    dec = [i for i, (k, _) in enumerate(seq) if k.startswith("conv3x3_in_kernel<4>")][0]
    return seq[:dec], seq[dec:]


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    fu, fd = second_pass(fetch)
    wu, wd = second_pass(write)
    assert [k for k, _ in fu] == [k for k, _ in wu] and [k for k, _ in fd] == [k for k, _ in wd], "passes differ"
    out = collections.OrderedDict()
    for part, fs, ws in (("unet", fu, wu), ("decode", fd, wd)):
        for (k, f), (_, w) in zip(fs, ws):
            e = out.setdefault(k, {"unet": {"launches": 0, "bytes": 0.0, "fetch_bytes_x2": 0.0, "write_bytes": 0.0},
                                   "decode": {"launches": 0, "bytes": 0.0, "fetch_bytes_x2": 0.0, "write_bytes": 0.0}})[part]
            e["launches"] += 1
            e["fetch_bytes_x2"] += 2.0 * f * 1024.0
            e["write_bytes"] += w * 1024.0
            e["bytes"] = e["fetch_bytes_x2"] + e["write_bytes"]
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    json.dump({"resolution": res, "note": "HBM-side bytes per kernel over one UNet step (CFG batch 2) and one decode: 2 x FETCH_SIZE + WRITE_SIZE, "
               "separate rocprofv3 --pmc passes of `bench.py --pmc-workload`", "kernels": out}, open(sys.argv[4] if len(sys.argv) > 4 else "profiles/r03_traffic.json", "w"), indent=1)
    for k, e in sorted(out.items(), key=lambda kv: -(50 * kv[1]["unet"]["bytes"] + kv[1]["decode"]["bytes"]))[:12]:
        n = 50 * e["unet"]["launches"] + e["decode"]["launches"]
        print("%-60s %9.2f MB / launch (weighted 50 x UNet + decode, %d launches)" % (k, (50 * e["unet"]["bytes"] + e["decode"]["bytes"]) / n / 1e6, n))


if __name__ == "__main__":
    main()
