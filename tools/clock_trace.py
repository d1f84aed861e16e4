#!/usr/bin/env python3
"""Which clock (and board power) does the part hold under a kernel?  ("power-limited" as evidence.)

While a workload loops for SECONDS on the main stream, ONE probe wave on a second stream samples the shader clock every millisecond
(diinn_debug_clock_probe: d(s_memtime) / d(s_memrealtime) x 100 MHz -- MI355X_MICROARCH.md, DVFS give-back item 6), and a host
thread samples what the driver publishes (sysfs sclk level / hwmon power, else rocm-smi / amd-smi) every 50 ms.  Workloads:

  idle          nothing (the probe alone)
  bf16_c5       BASELINE config 5's bf16 decode (decode_bf16_coop8p_kernel: 256 persistent workgroups)
  f32_c2        the fp32 headline decode (decode_kernel)
  x3_c2         the split-bf16 decode
  pbf16_c5      precompute_P_bf16_wide_kernel at c5 (the bf16_full mode's hoisted conv)
  wino4_144     conv_wino4_kernel, 192 x 192 x 512 channels WITHOUT a workspace: 144 workgroups resident (one partly filled round)
  wino4_256     conv_wino4_kernel, 256 x 256 x 512 channels: 256 workgroups resident (one full round)
  wino4_254     conv_wino4_kernel, 64 x 1016 x 512 channels: 254 work items = one round that leaves the probe's CU free
  wino4_split   192 x 192 x 512 WITH the workspace: the last round split over all 256 compute units

A kernel that fills every CU's register file (decode_bf16_coop8p_kernel: 2 waves x 254 registers per SIMD; conv_wino4_kernel: 16
waves per CU) cannot share a CU with the probe wave, which is resident first: a grid of 256 such workgroups then runs 255 + 1 and
takes twice its time.  --ncu N (DIINN_DEBUG_NCU) sizes the persistent grids and the F(4x4) split for N compute units, so that with
N = 254 the workload runs as it does alone on a 254-CU part and the probe keeps a CU to itself.

usage: python tools/clock_trace.py [SECONDS=2.0] [--ncu N] [workload ...]       (report on stdout)"""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diinn_amd._native as N  # noqa: E402
import diinn_amd.decoder as D  # noqa: E402
import diinn_amd.modules as M  # noqa: E402
import diinn_amd.synth as synth  # noqa: E402

dev = torch.device("cuda:0")
lib = N.load()
ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def host_sample():
    """(sclk MHz or None, power W or None, source) from whatever this box lets an ordinary user read."""
    sclk = power = None
    src = []
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for ln in open(f):
                if "*" in ln:
                    sclk = float(ln.split(":")[1].strip().split("M")[0])
                    src.append("sysfs pp_dpm_sclk")
            break
        except Exception:
            pass
    for pat in ("power1_average", "power1_input"):
        for f in glob.glob(f"/sys/class/drm/card*/device/hwmon/hwmon*/{pat}"):
            try:
                power = float(open(f).read()) / 1e6
                src.append(f"hwmon {pat}")
                break
            except Exception:
                pass
        if power is not None:
            break
    return sclk, power, "+".join(src)


def smi_sample():
    for cmd in (["rocm-smi", "--showclocks", "--showpower", "--json"], ["amd-smi", "metric", "--clock", "--power", "--json"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=10)
            if r.returncode == 0 and r.stdout.strip():
                return cmd[0], r.stdout.strip()[:1500]
        except Exception:
            continue
    return None, None


def workloads():
    w = {}
    sd = synth.decoder_state_dict(123)
    packed = D.pack_state_dict(sd).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)

    def decode(h, wd, hu, wu, compute, what="decode"):
        """the decode kernel alone (P computed once), or the P kernel alone (what="p")"""
        feat = torch.randn((1, 64, h, wd), device=dev, generator=gen)
        P = torch.empty((1, h, wd, 1024), device=dev)
        out = torch.empty((1, 3, hu, wu), device=dev)
        comp = N.COMPUTE[compute]
        st0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(lib.diinn_precompute_P_ex(st0, ptr(feat), ptr(packed), ptr(P), 1, h, wd, 0, h, comp), "P")

        def run():
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            if what == "p":
                N.check(lib.diinn_precompute_P_ex(st, ptr(feat), ptr(packed), ptr(P), 1, h, wd, 0, h, comp), "P")
            else:
                N.check(lib.diinn_decode_band_ex(st, ptr(P), ptr(packed), ptr(out), 1, h, wd, hu, wu, 0, hu, N.SIN_DEFAULT, comp), "decode")
        return run
    w["bf16_c5"] = decode(720, 1280, 2376, 4224, "bf16")
    w["f32_c2"] = decode(256, 256, 1024, 1024, "f32")
    w["x3_c2"] = decode(256, 256, 1024, 1024, "bf16x3")
    w["pbf16_c5"] = decode(720, 1280, 2376, 4224, "bf16_full", what="p")

    def wino4(hw, with_ws, ww=None):
        cin = 512
        hw, wdt = (hw, hw) if ww is None else (hw, ww)
        return wino4_hw(hw, wdt, cin, with_ws)

    def wino4_hw(h, wd, cin, with_ws):
        x = torch.randn((1, cin, h, wd), device=dev, generator=gen)
        wt = torch.randn((64, cin, 3, 3), device=dev, generator=gen) / (cin * 9) ** 0.5
        pk = M.pack_conv_wino4(wt).to(dev)
        bias = torch.zeros(64, device=dev)
        out = torch.empty((1, 64, h, wd), device=dev)
        wsf = lib.diinn_conv_wino4_workspace_floats()
        ws = torch.zeros(wsf, device=dev) if with_ws else None

        def run():
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            N.check(lib.diinn_conv_wino4_ws(st, ptr(x), cin * h * wd, cin, ptr(pk), ptr(bias), None, 0, ptr(out), 64 * h * wd, 1, 1, h, wd,
                                            ptr(ws) if with_ws else None, wsf if with_ws else 0), "conv_wino4")
        return run
    w["wino4_144"] = wino4(192, False)
    w["wino4_256"] = wino4(256, False)
    w["wino4_254"] = wino4(64, False, 1016)                      # 254 work items: every CU but the probe's pair busy, one round
    w["wino4_split"] = wino4(192, True)
    w["idle"] = None
    return w


def trace(name, fn, seconds):
    n = int(seconds * 1000) + 400
    samples = torch.zeros(3 * n, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    host = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            host.append((time.perf_counter(),) + host_sample())
            time.sleep(0.05)
    if fn is not None:
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        N.check(lib.diinn_debug_clock_probe(C.c_void_p(side.cuda_stream), ptr(samples), n, 100000), "clock_probe")   # 1 ms per sample
    launches = 0
    time.sleep(0.15)                                             # 150 ms of idle in front: the trace shows the drop
    t_load0 = time.perf_counter()
    if fn is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t_load0 < seconds:
            for _ in range(8):
                fn()
            launches += 8
            torch.cuda.current_stream().synchronize()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / max(launches, 1)
    else:
        time.sleep(seconds)
        ms = 0.0
    t_load1 = time.perf_counter()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    s = samples.cpu().view(-1, 3).double()
    ghz = (s[:, 0] / s[:, 1] * 0.1).tolist()
    lo, hi = int((t_load0 - t0) * 1000) + 100, int((t_load1 - t0) * 1000) - 50     # samples well inside the loaded interval
    loaded = sorted(ghz[lo:hi]) if hi > lo + 10 else sorted(ghz)
    idle = sorted(ghz[10:120])
    med = lambda v: v[len(v) // 2]  # noqa: E731
    hs = [h for h in host if t_load0 + 0.3 < h[0] < t_load1 - 0.05]
    sclk = [h[1] for h in hs if h[1] is not None]
    pw = [h[2] for h in hs if h[2] is not None]
    print(f"{name:12s} {launches:6d} launches, {ms:8.4f} ms each | in-kernel shader clock: idle head {med(idle):.3f} GHz, under load "
          f"median {med(loaded):.3f} (p10 {loaded[len(loaded) // 10]:.3f}, p90 {loaded[len(loaded) * 9 // 10]:.3f}) GHz"
          + (f" | driver: sclk median {med(sorted(sclk)):.0f} MHz" if sclk else " | driver sclk: n/a")
          + (f", power median {med(sorted(pw)):.0f} W (max {max(pw):.0f})" if pw else ", power: n/a")
          + (f" [{hs[0][3]}]" if hs and hs[0][3] else ""))
    # a coarse timeline: mean clock per 100 ms
    line = " ".join(f"{sum(ghz[i:i + 100]) / 100:.2f}" for i in range(0, len(ghz) - 99, 100))
    print(f"{'':12s} clock per 100 ms (GHz): {line}")
    return med(loaded)


def main():
    args = sys.argv[1:]
    if "--ncu" in args:
        i = args.index("--ncu")
        N.debug_set("DIINN_DEBUG_NCU", int(args[i + 1]))
        print(f"# DIINN_DEBUG_NCU = {args[i + 1]}")
        del args[i:i + 2]
    seconds = float(args[0]) if args and args[0].replace(".", "").isdigit() else 2.0
    names = [a for a in args if not a.replace(".", "").isdigit()]
    w = workloads()
    names = names or ["idle", "f32_c2", "x3_c2", "bf16_c5", "pbf16_c5", "wino4_144", "wino4_256", "wino4_split"]
    print(f"# {torch.cuda.get_device_name(0)}; probe = 1 wave, 1 ms per sample (100,000 ticks of the 100 MHz realtime counter); "
          f"workload loops {seconds} s after 150 ms of idle")
    which, txt = smi_sample()
    print(f"# smi at start ({which}): {txt}")
    for nme in names:
        trace(nme, w[nme], seconds)
        time.sleep(0.5)
    which, txt = smi_sample()
    print(f"# smi at end ({which}): {txt}")


if __name__ == "__main__":
    main()
