"""pmc_summary.txt -> traffic.json: HBM bytes per launch for the rANS encode / decode kernels.
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes
(MI355X_MICROARCH.md, HBM section: "reports exactly 1/2 of the bytes of a wide coalesced streaming read"), so it is
doubled for both kernels, whose lanes read whole 128-byte lines (calibration on the encode kernel: 2 x FETCH_SIZE
= 1.01 x the 1 GiB it must read; while the decode kernel still read 64-byte blocks, r01_v4, its FETCH_SIZE equalled
the stream bytes at face value and halved when it switched to whole lines).  WRITE_SIZE is taken as is."""
import json, re, sys
vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+mean/dispatch=([0-9.e+]+)", line)
    if m:
        kern = "rans_encode" if "encode" in m.group(1) else "rans_decode" if "decode" in m.group(1) else None
        if kern:
            vals.setdefault(kern, {})[m.group(2)] = float(m.group(3))
out = {}
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        mult = 2  # both kernels read whole 128-byte lines since r01_v6
        out[k] = int(v["FETCH_SIZE"] * 1024 * mult + v["WRITE_SIZE"] * 1024)
        out[k + "_detail"] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"], "fetch_multiplier": mult,
                              "read_bytes": int(v["FETCH_SIZE"] * 1024 * mult), "write_bytes": int(v["WRITE_SIZE"] * 1024)}
out["workload"] = {"coder": "rans", "table": "t256", "chunks": 262144, "chunk_len": 4096}
out["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per dispatch, 1 GiB T256 batch"
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
