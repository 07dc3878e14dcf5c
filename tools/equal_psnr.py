#!/usr/bin/env python3
"""Bitrate of ks265enc at the reference's PSNR-Y, from the same-clip tables under profiles/ (linear interpolation of kbps over PSNR-Y between the two neighbouring QPs
of ks265enc; extrapolation is refused).  usage: tools/equal_psnr.py profiles/r02_same_clips.txt [profiles/r02_same_clips_final_ks265enc.txt]
The first file supplies the reference rows (and ks265enc rows unless a second file overrides them section by section)."""
import re
import sys


def parse(path):
    out, sec = {}, None
    for line in open(path):
        if line.startswith("## "):
            sec = line[3:].strip()
            out[sec] = {"ref": None, "ours": []}
        m = re.search(r"bitrate, psnr:\s*([\d.]+)\s+([\d.]+)", line)
        if not m or sec is None:
            continue
        kbps, psnr = float(m.group(1)), float(m.group(2))
        if line.startswith("reference"):
            if out[sec]["ref"] is None or kbps:                      # keep the last reference line (highest thread count listed)
                out[sec]["ref"] = (kbps, psnr)
        elif line.startswith("ks265enc"):
            out[sec]["ours"].append((kbps, psnr))
    return out


def main():
    base = parse(sys.argv[1])
    if len(sys.argv) > 2:
        for sec, v in parse(sys.argv[2]).items():
            if sec in base and v["ours"]:
                base[sec]["ours"] = v["ours"]
    for sec, v in base.items():
        if not v["ref"] or len(v["ours"]) < 2:
            continue
        rk, rp = v["ref"]
        pts = sorted(v["ours"], key=lambda t: t[1])
        lo = [p for p in pts if p[1] <= rp]
        hi = [p for p in pts if p[1] >= rp]
        if not lo or not hi:
            print(f"{sec}: reference {rk:.0f} kbps at {rp:.2f} dB lies outside ks265enc's QP range ({pts[0][1]:.2f} .. {pts[-1][1]:.2f} dB)")
            continue
        (k0, p0), (k1, p1) = lo[-1], hi[0]
        k = k0 if p1 == p0 else k0 + (rp - p0) / (p1 - p0) * (k1 - k0)
        print(f"{sec}: reference {rk:.0f} kbps at {rp:.2f} dB; ks265enc at that PSNR-Y {k:.0f} kbps = {k / rk:.2f} x")


if __name__ == "__main__":
    main()
