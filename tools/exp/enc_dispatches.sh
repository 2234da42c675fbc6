#!/bin/bash
# Per-dispatch durations of the HIP / MIOpen kernels of ONE steady-state encoder pass (run on the GPU box).
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/enc_prof
rocprofv3 --kernel-trace --stats -d /tmp/enc_prof -o enc -- python ${GRAFT_REPO_ROOT:-/root/repo}/tools/prof_encoder.py 5 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/enc_prof/**/*.db", recursive=True)[0])
rows = db.cursor().execute("select name,start,end from kernels order by start").fetchall()
wa = [i for i, r in enumerate(rows) if "wa_presplit" in r[0]]
first = [i for i in wa if all("wa_presplit" not in rows[j][0] for j in range(max(0, i - 300), i) if rows[i][1] - rows[j][1] < 2e6)]
a = first[-1]
while a > 0 and rows[a][1] - rows[a - 1][2] < 3e5 and a > first[-1] - 80: a -= 1
b = len(rows)
tot = 0.0
for n, s, e in rows[a:b]:
    tot += (e - s) / 1e3
    print("%8.1f us  %s" % ((e - s) / 1e3, n[:90]))
print("sum %.1f us over %d dispatches, span %.1f us" % (tot, b - a, (rows[b - 1][2] - rows[a][1]) / 1e3))
PY
