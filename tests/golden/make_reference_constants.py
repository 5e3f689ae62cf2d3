"""Extracts the DATA constants the reference hard-codes on this path (numbers, not code) from its source text and pins
them in reference_constants.json: the 32 x 16 MinION channel layout (as a count and a SHA-256 of the comma-joined
channel numbers, lib/seq_routines.py:588-671), the number of hourly bins (:99), the channel count (:741, :905), the
histogram specifications handed to plotly (:228, :234, :241) and the pass/fail constants (lib/fast5_reader.py:37, :136,
lib/seq_routines.py:855).  Runs only where /root/reference is mounted:  python tests/golden/make_reference_constants.py
The reference's code cannot be run here (Python 2, h5py / Biopython absent); reading its text is all this does."""
import hashlib
import json
import os
import re

REF = "/root/reference/lib"
HERE = os.path.dirname(os.path.abspath(__file__))


def cli_options(src):
    """The option table of lib/seqstats.py:86-116 as data: flags, default, choices, type, nargs of every add_argument."""
    opts = []
    for m in re.finditer(r"add_argument\(\s*'(-\w+)',\s*'(--\w+)'(.*?)\)\s*\n\s*(?=[fg]\.add_argument|\n|try:)", src, re.S):
        short, long_, rest = m.group(1), m.group(2), m.group(3)
        o = {"short": short, "long": long_}
        d = re.search(r"default\s*=\s*(\[[^\]]*\]|None|\d+|ldir)", rest)
        o["default"] = None if not d or d.group(1) == "None" else ("<tmpdir>" if d.group(1) == "ldir" else eval(d.group(1)))
        c = re.search(r"choices\s*=\s*(\[[^\]]*\]|range\(\d+, \d+\))", rest)
        o["choices"] = None if not c else [x for x in eval(c.group(1))]
        o["type"] = "int" if re.search(r"type\s*=\s*int", rest) else None
        n = re.search(r"nargs\s*=\s*(\d+|\"\?\")", rest)
        o["nargs"] = None if not n else (int(n.group(1)) if n.group(1).isdigit() else "?")
        opts.append(o)
    return opts


def main():
    sr = open(os.path.join(REF, "seq_routines.py")).read()
    f5 = open(os.path.join(REF, "fast5_reader.py")).read()
    lines = sr.split("\n")
    a = next(i for i, l in enumerate(lines) if "NanoLayout = [[" in l)
    b = next(i for i in range(a, len(lines)) if lines[i].rstrip().endswith("]]"))
    layout = re.findall(r"'(\d+)'", "\n".join(lines[a:b + 1]))
    out = {
        "_source": "rsemeraro/PyPore lib/seq_routines.py, lib/fast5_reader.py (text only; see make_reference_constants.py)",
        "nano_layout": {"rows": 32, "cols": 16, "count": len(layout),
                        "sha256_of_comma_joined": hashlib.sha256(",".join(layout).encode()).hexdigest(),
                        "first_row": layout[:16], "last_row": layout[-16:]},
        "n_hours": int(re.search(r"HoursDict = \{str\(il\): \[0, 0, 0\] for il in range\(0, (\d+)\)\}", sr).group(1)),
        "n_channels": int(re.search(r"ChannelDict = \{str\(il\): \(\) for il in range\(1, (\d+)\)\}", sr).group(1)) - 1,
        "length_hist": {"nbinsx": int(re.search(r"LengthHist = go\.Histogram\(x=LengthsVec, nbinsx=(\d+)", sr).group(1))},
        "quality_hist": dict(zip(("start", "end", "size"), map(int, re.search(
            r"QualHist = go\.Histogram\(x=QualsVec, nbinsx=\d+, xbins=dict\(start=(\d+), end=(\d+), size=(\d+)\)", sr).groups()))),
        "gc_hist": dict(zip(("start", "end", "size"), map(int, re.search(
            r"GCHist = go\.Histogram\(x=GCVec, nbinsx=\d+, xbins=dict\(start=(\d+), end=(\d+), size=(\d+)\)", sr).groups()))),
        "fail_below_q": float(re.search(r"if float\(qual\) < ([0-9.]+):", f5).group(1)),
        "fail_below_q_summary": float(re.search(r"FailedReads = sum\(float\(i\[2\]\) < ([0-9.]+) for", sr).group(1)),
        "fastq_kept_from_q": float(re.search(r"if float\(res\[0\]\[4\]\) >= ([0-9.]+):", f5).group(1)),
        "fail_path_default_q": float(re.search(r"QScore = ([0-9.]+)", f5).group(1)),
        "muxes": list(range(*map(int, re.search(
            r"MuxProductivity = \{str\(il\): \[\] for il in range\((\d+), (\d+)\)\}", f5).groups()))),
    }
    out["output_suffixes"] = sorted(set(re.findall(r"prefix \+ '([._A-Za-z]+)'", sr)))   # :276, :773, :799
    out["seqstats_cli"] = cli_options(open(os.path.join(REF, "seqstats.py")).read())
    out["default_output_dir"] = re.search(r"out_dir = '([A-Za-z_]+)'", open(os.path.join(REF, "seqstats.py")).read()).group(1)
    json.dump(out, open(os.path.join(HERE, "reference_constants.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
