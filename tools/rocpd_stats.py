"""Summarise a rocprofv3 (ROCm 7.x) rocpd sqlite database into a per-kernel stats table,
the same content `rocprofv3 --stats` prints as kernel_stats.csv in older releases.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r1_kernel_stats.md
    python tools/rocpd_stats.py x_results.db 40 --last-frame 5   # only the dispatches after the 6th-from-last
                                                                 # decoder_kernel launch: the steady-state frame
                                                                 # (5 ray-chunk launches), without MIOpen's find pass
    python tools/rocpd_stats.py x_results.db 40 --periods 3      # the last 3 full periods between decoder launches (training
                                                                 # iterations launch the decoder once each)
"""
import sqlite3
import sys


def main(path, top=40, last_frame=0, periods=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    if last_frame:
        dec = sorted(e for n, s, e in rows if "decoder_kernel" in n or "decoder_pp_kernel" in n)
        if len(dec) > last_frame:
            t0 = dec[-last_frame - 1]
            rows = [r for r in rows if r[1] > t0]
    if periods:  # exactly `periods` repetitions of a loop that launches the decoder once per repetition (training iterations): from
        # the (periods+1)-th from last decoder launch up to, not including, the last one — warm-up (MIOpen's solver search) left out
        dec = sorted(s for n, s, e in rows if "decoder_kernel" in n or "decoder_pp_kernel" in n)
        if len(dec) > periods:
            rows = [r for r in rows if dec[-periods - 1] <= r[1] < dec[-1]]
    agg = {}
    for name, s, e in rows:
        d = (e - s) / 1e3  # ns -> us
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# kernel stats from {path.split('/')[-1]}: {len(rows)} dispatches, {total / 1e3:.3f} ms total GPU kernel time\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {a[0]} | {a[1] / 1e3:.3f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[1] / total:.2f} |")


if __name__ == "__main__":
    lf = int(sys.argv[sys.argv.index("--last-frame") + 1]) if "--last-frame" in sys.argv else 0
    pe = int(sys.argv[sys.argv.index("--periods") + 1]) if "--periods" in sys.argv else 0
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 40, lf, pe)
