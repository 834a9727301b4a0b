"""kernel_trace.csv of tools/probe_overlap.py -> do the encoder and cp_copy of the two-stream phases overlap in time?

Prints, for the last dispatches of the run, start / end of every rans_encode / cp_copy launch relative to the first of
them, with the queue it ran on, and the overlap of each cp_copy with the encoder launch nearest to it."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0]
        if "rans_encode" in name or "cp_copy" in name:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "enc" if "rans_encode" in name else "cp",
                         r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tail = rows[-n_show:]
t0 = tail[0][0]
for s, e, k, q, st in tail:
    print(f"{k:4s} queue {q:>3s} stream {st:>3s}  start {(s - t0) / 1e3:10.1f} us  end {(e - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us")
# overlap statistics over the whole run: for every cp_copy, the time it shares with any encoder launch
encs = [(s, e) for s, e, k, _, _ in rows if k == "enc"]
tot_cp = tot_ov = 0
for s, e, k, _, _ in rows:
    if k != "cp":
        continue
    tot_cp += e - s
    for es, ee in encs:
        tot_ov += max(0, min(e, ee) - max(s, es))
print(f"cp_copy time {tot_cp / 1e6:.3f} ms over the run, of which {tot_ov / 1e6:.3f} ms under an encoder launch")
