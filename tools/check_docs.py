"""Do the documents still say what the committed evidence says?   python tools/check_docs.py   (exit 1 and a list on any mismatch)

Figures quoted in profiles/README.md, DESIGN.md and EXPERIMENTS.md carry an invisible tag right behind them naming the file they come from:
    <!--chk csv=r05_kernel_stats_bench.csv name="k_skinny<2, 6, 4, 2, 0," avg_us=58.8-->   mean duration of the CSV row whose Name contains `name`
    <!--chk csv=... name="..." calls=1600-->                                                  its call count
    <!--chk log=r05_gpu_suite.log passed=206-->                                               pytest's "N passed" line
    <!--chk json=r05_bench_default.json key=ms_per_step value=20.39-->                        a (dotted) key of a one-line JSON; `scale` multiplies the file's number first
A tagged figure passes when it equals the file's value rounded to the digits the tag shows.  Also checked without tags: the stage table
of DESIGN section 7b and the newest section of profiles/README.md name the NEWEST round's kernel-stats CSV, and include/nested_diffusion.h
describes the operand storage the library really uses.  Run from the CPU suite (tests/test_abi_and_host.py)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
DOCS = ("profiles/README.md", "DESIGN.md", "EXPERIMENTS.md")      # the documents whose tagged figures are checked (and --fix rewrites)
TAG = re.compile(r"<!--chk\s+(.*?)-->")
KV = re.compile(r'(\w+)=("([^"]*)"|\S+)')


def _digits(s: str) -> int:
    return len(s.split(".")[1]) if "." in s else 0


def _same(quoted: str, actual: float) -> bool:
    return abs(round(actual, _digits(quoted)) - float(quoted)) < 0.5 * 10 ** (-_digits(quoted)) + 1e-12


def check_tag(args: dict) -> str:
    if "csv" in args:
        path = os.path.join(PROF, args["csv"])
        rows = [r for r in csv.DictReader(open(path)) if args["name"] in r["Name"]]
        if len(rows) != 1:
            return f"{args['csv']}: {len(rows)} rows match {args['name']!r}"
        r = rows[0]
        if "avg_us" in args and not _same(args["avg_us"], float(r["AverageNs"]) / 1e3):
            return f"{args['csv']} {args['name']!r}: quoted {args['avg_us']} us, file has {float(r['AverageNs']) / 1e3:.2f}"
        if "calls" in args and int(args["calls"]) != int(r["Calls"]):
            return f"{args['csv']} {args['name']!r}: quoted {args['calls']} calls, file has {r['Calls']}"
        return ""
    if "log" in args:
        m = re.search(r"(\d+) passed", open(os.path.join(PROF, args["log"])).read())
        if not m or int(m.group(1)) != int(args["passed"]):
            return f"{args['log']}: quoted {args['passed']} passed, file has {m.group(1) if m else 'no such line'}"
        return ""
    if "json" in args:
        txt = [l for l in open(os.path.join(PROF, args["json"])).read().splitlines() if l.strip().startswith("{")][-1]
        v = json.loads(txt)
        for k in args["key"].split("."):
            v = v[k]
        v = float(v) * float(args.get("scale", "1"))
        if not _same(args["value"], v):
            return f"{args['json']} {args['key']}: quoted {args['value']}, file has {v:.4g}"
        return ""
    return f"unknown tag {args}"


def actual_value(args: dict):
    """(field name in the tag, value in the file) of a tag, for --fix"""
    if "csv" in args:
        rows = [r for r in csv.DictReader(open(os.path.join(PROF, args["csv"]))) if args["name"] in r["Name"]]
        if len(rows) != 1:
            return None
        return ("avg_us", float(rows[0]["AverageNs"]) / 1e3) if "avg_us" in args else ("calls", int(rows[0]["Calls"]))
    if "log" in args:
        m = re.search(r"(\d+) passed", open(os.path.join(PROF, args["log"])).read())
        return ("passed", int(m.group(1))) if m else None
    if "json" in args:
        txt = [l for l in open(os.path.join(PROF, args["json"])).read().splitlines() if l.strip().startswith("{")][-1]
        v = json.loads(txt)
        for k in args["key"].split("."):
            v = v[k]
        return ("value", float(v) * float(args.get("scale", "1")))
    return None


def fix() -> int:
    """--fix: rewrite every tagged figure (the number right in front of its tag, and the tag's own copy) from the file the tag names, at
    the digits the tag shows.  For re-profiled rounds: the prose around the figures is the author's business, the figures are the files'."""
    n = 0
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        text = open(path).read()

        def sub(m):
            nonlocal n
            args = {k: (q if q is not None and v.startswith('"') else v) for k, v, q in KV.findall(m.group(2))}
            try:
                got = actual_value(args)
            except Exception:
                got = None
            if got is None:
                return m.group(0)
            field, val = got
            new = f"{val:.{_digits(args[field])}f}" if not isinstance(val, int) else str(val)
            if new != args[field]:
                n += 1
            tag = re.sub(rf"\b{field}={re.escape(args[field])}", f"{field}={new}", m.group(2))
            return f"{new}<!--chk {tag}-->"

        text2 = re.sub(r"([0-9][0-9.]*)<!--chk\s+(.*?)-->", sub, text)
        if text2 != text:
            open(path, "w").write(text2)
    print(f"check_docs --fix: {n} figure(s) rewritten")
    return 0


def main() -> int:
    if "--fix" in sys.argv[1:]:
        return fix()
    problems, n_tags = [], 0
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for m in TAG.finditer(text):
            args = {k: (q if q is not None and v.startswith('"') else v) for k, v, q in KV.findall(m.group(1))}
            n_tags += 1
            try:
                err = check_tag(args)
            except Exception as e:                                   # a missing file is a finding, not a crash
                err = f"{type(e).__name__}: {e}"
            if err:
                problems.append(f"{doc}: {err}")
    rounds = sorted({int(m.group(1)) for f in os.listdir(PROF) for m in [re.match(r"r(\d\d)_kernel_stats_bench\.csv$", f)] if m})
    newest = f"r{rounds[-1]:02d}"
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = design[design.index("## 7b."):]
    sec = sec[:sec.index("\n## ", 4)] if "\n## " in sec[4:] else sec
    table_intro = sec[:sec.index("| stage |")]
    if f"profiles/{newest}_kernel_stats_bench.csv" not in table_intro:
        problems.append(f"DESIGN.md section 7b: the stage table does not cite profiles/{newest}_kernel_stats_bench.csv")
    readme = open(os.path.join(PROF, "README.md")).read()
    first = readme[readme.index("## round"):]
    first = first[:first.index("\n## round", 8)] if "\n## round" in first[8:] else first
    if f"`{newest}_kernel_stats_bench.csv`" not in first or f"## round {rounds[-1]} " not in first[:40] + " ":
        problems.append(f"profiles/README.md: the first section is not round {rounds[-1]} or does not list {newest}_kernel_stats_bench.csv")
    n_design = len(design.splitlines())
    if n_design > 420:
        problems.append(f"DESIGN.md has grown to {n_design} lines: measured-and-removed material belongs in EXPERIMENTS.md")
    if n_tags < 8:
        problems.append(f"only {n_tags} <!--chk ...--> tags found: the quoted figures are no longer tied to their files")
    header = open(os.path.join(ROOT, "include", "nested_diffusion.h")).read()
    conv = header[:header.index("#ifndef NESTED_DIFFUSION_H")]
    if "f32-input MFMA, exact f32 products/accumulate" in conv or "frag32b3" not in conv:
        problems.append("include/nested_diffusion.h: the convention block still describes fp32 storage / f32-input MFMAs throughout")
    for p in problems:
        print("MISMATCH", p)
    print(f"check_docs: {n_tags} tagged figures, {len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
