"""Wave-state breakdown of the hot kernels from a rocprofv3 --pmc pass with
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE:
    python tools/pmc_wave_states.py <dir with the rocpd .db>
WAIT_ANY = parked in s_waitcnt / s_barrier, WAIT_INST_ANY = issue stall (mostly: the matrix pipe is busy with the other
wave), MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)."""
import glob
import re
import sqlite3
import sys

for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    rows = con.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value) from pmc_events "
                       "group by name, counter_name").fetchall()
    ker = {}
    for n, c, cnt, tot, avg in rows:
        k = re.sub(r"\(anonymous namespace\)::", "", n)
        k = re.sub(r"\(.*$", "", k)
        ker.setdefault(k, {})[c] = (cnt, tot, avg)
    for k, v in sorted(ker.items()):
        if not any(x in k for x in ("half1", "half2", "dgemm_tn_acc_dma")):
            continue
        wc = v["SQ_WAVE_CYCLES"][1]
        print(k)
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in v:
                print("   %-22s %5.1f %% of wave cycles" % (c, 100.0 * v[c][1] / wc))
        gui, mf = v["GRBM_GUI_ACTIVE"], v["SQ_VALU_MFMA_BUSY_CYCLES"]
        # per dispatch: GUI_ACTIVE is reported per XCD (8 rows), the SQ counters per shader engine (32 rows)
        ndisp = gui[0] / 8.0
        print("   dispatches %d: MFMA busy %.1f %% of the SIMD cycles, wave cycles / (GUI x 1024 SIMDs x 4) = %.2f waves per SIMD"
              % (ndisp, 100.0 * mf[1] / (gui[2] * 1024.0 * ndisp), wc * 4.0 / (gui[2] * 1024.0 * ndisp)))
