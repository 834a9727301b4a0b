"""pmc_summary.txt + the bench.py line of the same run -> one traffic entry: HBM bytes per launch of the encode and the
decode kernel of that workload (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, mean per dispatch).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte read request at 64 bytes
(MI355X_MICROARCH.md, HBM section: "reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Calibration used here, per kernel:
the bytes the kernel MUST read are known (encode: the symbols; decode: the streams); if the counter at face value is below
0.75 x that minimum, the kernel's reads are of the wide kind and the counter is doubled, else it is taken as is.  Both the
raw value and the multiplier are recorded.  WRITE_SIZE is taken as is (it matched the stream bytes within 1-2 % on every
kernel of this library that was checked by construction).

usage: make_traffic_json.py pmc_summary.txt bench.json out_entry.json     (one entry)
       make_traffic_json.py --merge profiles/traffic.json entry.json ...  (replace entries with the same key)"""
import json, re, sys


def entry(pmc_path, bench_path, out_path):
    b = json.loads([ln for ln in open(bench_path) if ln.startswith("{")][0])
    vals = {}
    for line in open(pmc_path):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+mean/dispatch=([0-9.e+]+)", line)
        if m:
            # cp_copy = the compaction pass of roofline_dense (the three scan kernels next to it move < 0.5 % of its bytes)
            side = ("compact" if "cp_copy" in m.group(1) else None if "cp_" in m.group(1) else
                    "encode" if "encode" in m.group(1) else "decode" if "decode" in m.group(1) else None)
            if side:
                vals.setdefault(side, {"kernel": m.group(1).strip()})[m.group(2)] = float(m.group(3))
    in_bytes = b["config"]["chunks_per_gpu"] * b["config"]["chunk_len"]
    stream_bytes = b["roofline_encode"]["algorithmic_bytes_per_launch"] - in_bytes
    # csrc_sha: the kernel sources the pass was taken on (bench.py csrc_sha()); bench.py quotes an entry only for the same
    out = {"key": b["traffic_key"], "csrc_sha": b.get("csrc_sha"), "core_sha": b.get("core_sha"),
           "algorithmic_bytes_per_launch": b["roofline_encode"]["algorithmic_bytes_per_launch"]}
    for side, v in vals.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        must_read = in_bytes if side == "encode" else stream_bytes  # (compact: the streams, out of their slots)
        raw = v["FETCH_SIZE"] * 1024
        mult = 2 if raw < 0.75 * must_read else 1
        out[side] = int(raw * mult + v["WRITE_SIZE"] * 1024)
        out[side + "_detail"] = {"kernel": v["kernel"], "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                                 "fetch_multiplier": mult, "bytes_the_kernel_must_read": int(must_read),
                                 "read_bytes": int(raw * mult), "write_bytes": int(v["WRITE_SIZE"] * 1024)}
    out["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per dispatch"
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out))


def merge(target, entries):
    try:
        cur = json.load(open(target))
    except Exception:
        cur = {}
    lst = cur.get("entries", [])
    for p in entries:
        e = json.load(open(p))
        lst = [x for x in lst if x.get("key") != e["key"]] + [e]
    json.dump({"entries": lst, "note": "one entry per bench.py workload (key = bench.py's traffic_key); see tools/make_traffic_json.py"},
              open(target, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "--merge":
        merge(sys.argv[2], sys.argv[3:])
    else:
        entry(*sys.argv[1:4])
