#!/bin/bash
# Kernel launches of the LAST repetition of a command, in launch order with durations (run ON THE GPU BOX):
#   tools/seq_trace.sh <out dir> <marker kernel substring> <command...>
# rocprofv3 --kernel-trace over the command; the launches after the last occurrence of a kernel whose name contains the marker
# markers: comma-separated; each gets sequence_<marker>.txt, ending at the next launch of any marker
# (e.g. ds_mel_to_cl = first kernel of a vocode, ds_codebook_gather = first kernel of a decode) are listed.
set -u
OUT=$1; MARK=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/trace" -o t --output-format csv -- "$@" > "$OUT/cmd.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python - "$F" "$MARK" "$OUT/sequence.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def emit(seq, out):
    t0 = int(seq[0]["Start_Timestamp"])
    tot = 0.0
    for r in seq:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        gx = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        wx = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
        out.write("%9.1f us  +%8.1f us  grid %-9s wg %-4s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, gx, wx,
                                                                  r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]))
    wall = (int(seq[-1]["End_Timestamp"]) - t0) / 1e3
    line = "%d launches, kernels %.1f us, wall %.1f us" % (len(seq), tot, wall)
    out.write(line + "\n")
    print(line)


marks = sys.argv[2].split(",")
anym = [i for i, r in enumerate(rows) if any(m in r["Kernel_Name"] for m in marks)]
for mk in marks:
  mine = [i for i in anym if mk in rows[i]["Kernel_Name"]]
  if not mine:
    continue
  nxt = [i for i in anym if i > mine[-1]]
  seq = rows[mine[-1]:(nxt[0] if nxt else len(rows))]
  out = open(sys.argv[3].replace(".txt", "_%s.txt" % mk), "w")
  emit(seq, out)
PY
rm -rf "$OUT/trace"
