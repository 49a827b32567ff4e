"""Per-phase timeline of the stage-1 kernels from the s_memtime stamps of sjhip_stage1_trace (gpurun_out/s1_trace_v*.npz,
written by tools/s1_experiment.py).  The counters of different XCDs are not synchronised, so every statistic is taken
inside one workgroup (identified by XCC_ID and the CU / SE bits of HW_ID): for each wave of each workgroup the time
between its first phase-A stamp and its last flatten stamp is split into phase A, flatten and the rest (= waiting: for
the other waves at a barrier / for the result record, for the serial section, inside the serial section).
usage: python tools/s1_timeline.py gpurun_out/s1_trace_v1.npz [more.npz ...]  -> JSON on stdout"""
import json
import sys

import numpy as np


def analyse(path):
    t = np.load(path)["trace"].astype(np.int64)
    tiles, waves, _ = t.shape
    hw = t[:, 0, 5]
    key = ((hw >> 32) << 32) | (hw & 0xFFFFFF00 & ~0x3F)  # XCC | HW_ID without wave / SIMD bits
    blocks = {}
    for ti in range(tiles):
        blocks.setdefault(int(key[ti]), []).append(ti)
    a_tot = f_tot = span_tot = 0
    serial, rounds, a_spread, a_dur, f_dur, res_wait = [], [], [], [], [], []
    for tl in blocks.values():
        tl = sorted(tl, key=lambda x: t[x, 0, 0])
        if len(tl) < 3:
            continue
        sub = t[tl]  # [k, wave, word] in time order
        first = sub[0, :, 0]
        last = sub[-1, :, 4]
        span_tot += int((last - first).sum())
        a = sub[:, :, 1] - sub[:, :, 0]
        f = sub[:, :, 4] - sub[:, :, 3]
        a_tot += int(a.sum())
        f_tot += int(f.sum())
        a_dur += a[1:-1].ravel().tolist()
        f_dur += f[1:-1].ravel().tolist()
        a_spread += (sub[1:-1, :, 1].max(1) - sub[1:-1, :, 1].min(1)).tolist()
        rounds += np.diff(sub[:, 0, 3]).tolist()
        # the serial section of tile k: from the moment the whole block knows tile k+1 is aggregated (last phase-A end
        # of tile k+1 in the barrier kernels; unknown who ran it in the stealing kernel) to its end stamp
        s_end = sub[:, :, 2].max(1)
        for k in range(1, len(tl) - 1):
            if s_end[k] > 0:
                res_wait.append(int(sub[k, :, 3].min() - s_end[k]))
        if True:
            for k in range(1, len(tl) - 1):
                w = int(np.argmax(sub[k, :, 2]))
                if sub[k, w, 2] > 0:
                    # the wave that ran the duty: from its previous stamp to the end of the duty
                    prev = max(int(sub[k + 1, w, 1]) if sub[k + 1, w, 1] < sub[k, w, 2] else 0,
                               int(sub[k - 1, w, 4]) if sub[k - 1, w, 4] < sub[k, w, 2] else 0,
                               int(sub[k, w, 1]))
                    serial.append(int(sub[k, w, 2]) - prev)
    q = lambda v, p: float(np.percentile(v, p)) if len(v) else None
    return {"file": path, "tiles": tiles, "waves_per_block": waves, "blocks": len(blocks),
            "tile_round_cycles_median": q(rounds, 50),
            "phaseA_cycles": {"p5": q(a_dur, 5), "median": q(a_dur, 50), "p95": q(a_dur, 95)},
            "phaseA_end_spread_inside_a_block_median": q(a_spread, 50),
            "flatten_cycles": {"median": q(f_dur, 50), "p95": q(f_dur, 95)},
            "serial_duty_cycles": {"median": q(serial, 50), "p95": q(serial, 95)},
            "fraction_of_wave_time": {"phaseA": round(a_tot / span_tot, 4), "flatten": round(f_tot / span_tot, 4),
                                      "waiting_or_serial": round(1 - (a_tot + f_tot) / span_tot, 4)}}


if __name__ == "__main__":
    print(json.dumps([analyse(p) for p in sys.argv[1:]], indent=1))
