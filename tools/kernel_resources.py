"""Per-kernel register / scratch / occupancy table of libqf_hip.so from hipcc's -Rpass-analysis=kernel-resource-usage.

    QF_EXTRA_HIPCC_FLAGS="-Rpass-analysis=kernel-resource-usage" python -m quadraturefields_amd.build --force > /tmp/res.txt 2>&1
    python tools/kernel_resources.py /tmp/res.txt > profiles/r3/kernel_resources.md
"""
import re
import subprocess
import sys


def main(path):
    out, cur = [], None
    for line in open(path).read().splitlines():
        m = re.search(r"remark: (.*) \[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = dem.replace("(anonymous namespace)::", "")
            dem = re.sub(r"^void ", "", dem)
            depth, cut = 0, len(dem)
            for i, ch in enumerate(dem):          # cut at the argument list, keep template arguments
                if ch == "<":
                    depth += 1
                elif ch == ">":
                    depth -= 1
                elif ch == "(" and depth == 0:
                    cut = i
                    break
            cur = {"name": dem[:cut]}
            out.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    seen = set()
    print("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950: every kernel of libqf_hip.so\n")
    print("kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | occupancy waves/SIMD | LDS B/block")
    print("---|---|---|---|---|---|---")
    for o in out:
        if o["name"] in seen:
            continue
        seen.add(o["name"])
        print(f"{o['name']} | {o.get('VGPRs')} | {o.get('AGPRs')} | {o.get('SGPRs')} | {o.get('ScratchSize [bytes/lane]')} | "
              f"{o.get('Occupancy [waves/SIMD]')} | {o.get('LDS Size [bytes/block]')}")


if __name__ == "__main__":
    main(sys.argv[1])
