#!/usr/bin/env python3
"""The table of profiles/README.md that prices the committed rocprofv3 summaries: for every `<tag>_rocprofv3_kernel_stats_<name>_depth1.csv`
(`rocprofv3 --kernel-trace --stats` of `bench.py ... --passes-per-step 1 --depth 1`: one pass in flight, nothing overlaps) the
dominant pixel kernel's AverageNs as the file holds it, the bytes one such launch must move (bench.compulsory_bytes: the same
model bench.py's `roofline` object uses) and the resulting fraction of the 8 TB/s peak.

    python tools/profiles_table.py r06            print the markdown table
    python tools/profiles_table.py r06 --write    rewrite it between the markers in profiles/README.md

tests/test_host_and_abi.py::test_profiles_readme_table_matches_the_tracked_csv recomputes it and compares it with the prose
(the README quoted 1 460 617 ns for a CSV that held 1 504 821)."""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# name in the file name -> (H, W, n, frames per launch, dense output, detector, what ran)
RUNS = {
    "C2": (1080, 1920, 8, 4000, False, "m3", "1920x1080, n = 8 (headline)"),
    "C3": (2160, 3840, 16, 1000, False, "m3", "3840x2160, n = 16, mask-east"),
    "C5": (4320, 7680, 8, 500, True, "m3", "7680x4320, n = 8, m3det_high, dense output format"),
    "C1s": (540, 960, 8, 16000, False, "m3", "960x540, n = 8"),
    "C2n25": (1080, 1920, 25, 4000, False, "m3", "1920x1080, n = 25 (the reference's default window_sec 1 at 25 fps)"),
    "C1sn25": (540, 960, 25, 16000, False, "m3", "960x540, n = 25"),
    "classic": (1080, 1920, 4, 4000, False, "classic", "ClassicDetector, 1920x1080, sparse format"),
}
BEGIN, END = "<!-- profiles_table:begin -->", "<!-- profiles_table:end -->"


def rows(tag):
    out = []
    for name, (H, W, n, B, dense, det, what) in RUNS.items():
        path = os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_kernel_stats_{name}_depth1.csv")
        if not os.path.exists(path):
            continue
        want = "k_classic_diff" if det == "classic" else "k_stack_threshold"
        best = None
        for r in csv.DictReader(open(path)):
            if want in r["Name"] and (best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                best = r
        if best is None:
            continue
        avg = float(best["AverageNs"])
        nbytes, chunks, _ = bench.compulsory_bytes(H, W, n, B, dense, det)
        out.append((os.path.basename(path), best["Name"].split("(")[0], int(best["Calls"]), avg, nbytes, chunks, nbytes / avg / 8000.0, what))
    return out


def table(tag):
    lines = ["| file | kernel | launches | AverageNs | bytes per launch (time chunks) | fraction of 8 TB/s | workload |", "|---|---|---|---|---|---|---|"]
    for f, k, calls, avg, nb, ch, frac, what in rows(tag):
        lines.append(f"| `{f}` | `{k}` | {calls} | {avg:.0f} | {nb} ({ch}) | {frac:.3f} | {what} |")
    return "\n".join(lines)


def parse(text):
    """[(file, AverageNs, bytes, frac)] of the table between the markers."""
    body = text[text.index(BEGIN) + len(BEGIN):text.index(END)]
    out = []
    for ln in body.splitlines():
        m = re.match(r"\| `([^`]+)` \| `[^`]+` \| \d+ \| (\d+) \| (\d+) \(\d+\) \| ([0-9.]+) \|", ln)
        if m:
            out.append((m.group(1), float(m.group(2)), int(m.group(3)), float(m.group(4))))
    return out


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    t = table(tag)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "profiles", "README.md")
        text = open(p).read()
        text = text[:text.index(BEGIN) + len(BEGIN)] + "\n" + t + "\n" + text[text.index(END):]
        open(p, "w").write(text)
    print(t)
