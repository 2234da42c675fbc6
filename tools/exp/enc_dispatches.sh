#!/bin/bash
# Per-dispatch durations of the HIP / MIOpen kernels of ONE steady-state encoder pass (run on the GPU box).
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/enc_prof
rocprofv3 --kernel-trace --stats -d /tmp/enc_prof -o enc -- python ${GRAFT_REPO_ROOT:-/root/repo}/tools/prof_encoder.py 5 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/enc_prof/**/*.db", recursive=True)[0])
rows = db.cursor().execute("select name,start,end from kernels order by start").fetchall()
stem = [i for i, r in enumerate(rows) if "conv_stem_kernel" in r[0]]
a, b = stem[-2], stem[-1]   # one steady-state pass: from one stem convolution to the next
tot = 0.0
for n, s, e in rows[a:b]:
    tot += (e - s) / 1e3
    print("%8.1f us  %s" % ((e - s) / 1e3, n[:90]))
print("sum %.1f us over %d dispatches, span %.1f us" % (tot, b - a, (rows[b - 1][2] - rows[a][1]) / 1e3))
PY
