#!/usr/bin/env python3
"""clock_probe.py — what the chip's clock and power do under the Flat scan kernels ("the wide tile sits on the CU's vector-memory path under a
clock the power governor lowers" was an inference; this is the measurement). A thread samples the GPU's sysfs sensors (hwmon freq1_input = shader clock,
power1_average / power1_input = package power) every ~2 ms while one kernel at a time runs back to back for a few seconds:
  * flat cosine 1M x 768, B = 256  (flat_scan_qr<0>: the wide tile, int8 MFMA + LDS-DMA ring)
  * flat L2^2   1M x 768, B = 256  (flat_scan_qr<1>)
  * flat L2^2   1M x 768, B = 64   (flat_scan_qn<1>: the narrow tile, the HBM-bound regime)
  * a plain streaming read (torch: sum over 3 GB)
  * idle
For each: kernel ms (HIP events on the kernel's own dispatch), mean / min / max shader clock and power over the leg's samples.
usage: clock_probe.py [rows]"""
import glob
import json
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import comet_amd as ca  # noqa: E402


def sensor_files(only=None):
    """hwmon sensor files of one card (`only`), or {card: files} of every card of the host (the box shows all of its GPUs in sysfs; ours is the one whose clock moves)"""
    if only is None:
        cards = {}
        for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            f = sensor_files(h)
            if "sclk_hz" in f:
                cards[h] = f
        return cards
    out = {}
    for h in (only,):
        for key, names in (("sclk_hz", ("freq1_input",)), ("power_uw", ("power1_average", "power1_input")), ("mclk_hz", ("freq2_input",)), ("temp_mc", ("temp1_input",))):
            for n in names:
                p = Path(h) / n
                if p.exists() and key not in out:
                    try:
                        int(p.read_text().strip()); out[key] = p
                    except Exception:
                        pass
        if "sclk_hz" in out:
            break
    return out


class Sampler(threading.Thread):
    def __init__(self, files, period=0.002):
        super().__init__(daemon=True)
        self.files, self.period, self.rows, self.on = files, period, [], True

    def run(self):
        while self.on:
            t = time.perf_counter()
            r = {"t": t}
            for k, p in self.files.items():
                try:
                    r[k] = int(p.read_text().strip())
                except Exception:
                    pass
            self.rows.append(r)
            dt = self.period - (time.perf_counter() - t)
            if dt > 0:
                time.sleep(dt)


def stats(rows, key, scale):
    v = [r[key] * scale for r in rows if key in r]
    if not v:
        return None
    v.sort()
    return {"mean": round(sum(v) / len(v), 1), "min": round(v[0], 1), "p10": round(v[len(v) // 10], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1), "samples": len(v)}


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = 768
    cards = sensor_files()
    ctx = ca.Context(0)
    ctx.set_lanes(1)
    fill = lambda buf, lo, m: ctx.synth_fill(buf, 0xC0FFEE + 1, lo * d, m * d)
    import bench
    legs = []
    idx = {}
    for name, metric in (("cosine", ca.COSINE), ("l2", ca.L2_SQUARED)):
        g = ca.FlatIndex(ctx, d, metric)
        bench.add_rows(ctx, g, 0, rows, d, fill)
        idx[name] = g
    q = ctx.alloc(256 * d * 4); ctx.synth_fill(q, 0xBEEF + 1, 0, 256 * d)
    o = (ctx.alloc(256 * 100 * 4), ctx.alloc(256 * 100 * 4), ctx.alloc(256 * 4))
    import torch
    big = torch.ones(int(1.5e9), dtype=torch.float16, device="cuda")

    def flat(name, B):
        def f():
            idx[name].search_batch_dev(q, B, 100, *o, 100, mode=2)
        return f

    plan = [("idle", None, None), ("flat cosine B 256 (flat_scan_qr<0>, wide tile)", flat("cosine", 256), "flat_scan_i8"), ("flat L2^2 B 256 (flat_scan_qr<1>, wide tile)", flat("l2", 256), "flat_scan_i8"),
            ("flat L2^2 B 64 (flat_scan_qn<1>, narrow tile)", flat("l2", 64), "flat_scan_i8_n64"), ("streaming read 3 GB (torch sum)", lambda: big.sum(dtype=torch.float32), None), ("idle again", None, None)]
    # which card is ours: the one whose shader clock is highest while the wide tile runs
    t_end = time.perf_counter() + 1.0
    peak = {h: 0 for h in cards}
    while time.perf_counter() < t_end:
        for _ in range(20):
            plan[1][1]()
        for h, f in cards.items():
            try:
                peak[h] = max(peak[h], int(f["sclk_hz"].read_text().strip()))
            except Exception:
                pass
    ctx.sync()
    mine = max(peak, key=peak.get)
    files = cards[mine]
    print("cards:", {h: round(v * 1e-6) for h, v in peak.items()}, "-> sampling", mine, flush=True)
    time.sleep(1.0)
    out = []
    for name, fn, kern in plan:
        if fn:
            for _ in range(20):
                fn()
            ctx.sync(); torch.cuda.synchronize()
        s = Sampler(files); s.start()
        t0 = time.perf_counter(); n = 0
        if fn:
            ctx.profile(True); ctx.profile_reset()
            while time.perf_counter() - t0 < 3.0:
                for _ in range(50):
                    fn()
                n += 50
                ctx.sync(); torch.cuda.synchronize()
        else:
            time.sleep(1.5)
        wall = time.perf_counter() - t0
        s.on = False; s.join()
        rec = {"leg": name, "calls": n, "ms_per_call_wall": round(wall / n * 1e3, 4) if n else None}
        if fn:
            p = ctx.profile_dump(); ctx.profile(False)
            if kern and kern in p:
                rec["kernel"] = kern; rec["kernel_ms"] = round(p[kern][0] / max(1, p[kern][1]), 4)
            rec["kernels_seen"] = {k: round(v[0] / max(1, v[1]), 4) for k, v in p.items()}
        body = s.rows[len(s.rows) // 10:]                       # the first tenth: the governor's ramp
        rec["sclk_mhz"] = stats(body, "sclk_hz", 1e-6)
        rec["power_w"] = stats(body, "power_uw", 1e-6)
        rec["mclk_mhz"] = stats(body, "mclk_hz", 1e-6)
        rec["temp_c"] = stats(body, "temp_mc", 1e-3)
        out.append(rec)
        print(json.dumps(rec), flush=True)
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/clock_probe.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
