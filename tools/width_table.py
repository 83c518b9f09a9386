"""profiles/r03_width_table.md from profiles/r03_width_ab.txt and profiles/r03_width_pmc.txt (both written by tools/profile_r03.sh).
usage: python tools/width_table.py        (prints the table and rewrites the file)"""
import collections
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ab = open(os.path.join(ROOT, "profiles/r03_width_ab.txt")).read()
    rows = collections.OrderedDict()
    for m in re.finditer(r"^(\w+) width (\d): ([\d.]+) Mrays/s; one dispatch ([\d.]+) kernel ms ([\d.]+) nodes/ray ([\d.]+) prims/ray ([\d.]+)", ab, re.M):
        r = rows.setdefault((m.group(1), int(m.group(2))), dict(one=[], rays=[], nodes=float(m.group(6)), prims=float(m.group(7))))
        r["one"].append(float(m.group(4))); r["rays"].append(float(m.group(4)) * float(m.group(5)) * 1e3)
    ratio = {}
    for sec in re.split(r"^== ", ab, flags=re.M)[1:]:
        name = sec.split()[0]
        base = [float(x) for x in re.findall(r"^base ([\d.]+)", sec, re.M)]
        three = [float(x) for x in re.findall(r"^libptrace_3load.so ([\d.]+)", sec, re.M)]
        if base and three:
            ratio[name] = max(three) / max(base)
    cnt = {}
    for sec in re.split(r"^== ", open(os.path.join(ROOT, "profiles/r03_width_pmc.txt")).read(), flags=re.M)[1:]:
        p = sec.split("\n")[0].split()
        key = (p[0], int(p[2])) if p[1] == "width" else (p[0], p[1])
        cnt[key] = {m.group(1): float(m.group(2)) for m in re.finditer(r"(SQ_\w+)\s+dispatches=\s*\d+ sum=\S+ per_dispatch=(\S+)", sec)}
    out = ["Node widths side by side (profiles/r03_width_ab.txt, r03_width_pmc.txt; 1920x1080, 8 bounces, 16 spp - city10m 4 spp; the four-wide tree is the device builder's, "
           "the eight-wide tree the host builder's; the three-load row is a measurement build of the same four-wide tree, its rate the ratio to the installed library in the same run "
           "applied to the first row).", "",
           "| scene | node | node visits / ray | primitive tests / ray | vector instructions / ray | lanes / vector instruction | SQ_WAIT_ANY / SQ_WAVE_CYCLES | Mrays/s, one dispatch at a time |",
           "|---|---|---|---|---|---|---|---|"]

    def fmt(scene, label, r, d, one):
        if d:
            rays = max(r["rays"])
            vi, ln, w = "%.1f" % (d["SQ_INSTS_VALU"] / rays), "%.1f" % (d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"]), "%.2f" % (d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"])
        else:
            vi = ln = w = "-"
        return "| %s | %s | %.3f | %.3f | %s | %s | %s | %.0f |" % (scene, label, r["nodes"], r["prims"], vi, ln, w, one)

    for s in ("hall260k", "blob70k", "city10m"):
        r4, r8 = rows[(s, 4)], rows[(s, 8)]
        out.append(fmt(s, "four-wide (ships)", r4, cnt.get((s, 4)), max(r4["one"])))
        k3 = [k for k in cnt if k[0] == s and k[1] not in (4, 8)]
        out.append(fmt(s, "four-wide, 48-byte record (three loads)", r4, cnt[k3[0]] if k3 else None, max(r4["one"]) * ratio[s]))
        out.append(fmt(s, "eight-wide, octant order", r8, cnt.get((s, 8)), max(r8["one"])))
    text = "\n".join(out) + "\n"
    open(os.path.join(ROOT, "profiles/r03_width_table.md"), "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
