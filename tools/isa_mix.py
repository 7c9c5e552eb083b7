#!/usr/bin/env python3
"""Static instruction mix of one kernel (and of its hottest loop) from a gfx950 assembly listing.

    hipcc -O3 ... -S --cuda-device-only focnerf_amd/csrc/ffmlp.hip -o /tmp/ffmlp.s
    python tools/isa_mix.py /tmp/ffmlp.s '_Z15k_mlp_bwd_fusedILi64ELi3ELi1ELb1ELi3ELb1E' [--loop] [--dump out.s]

Prints registers / LDS from the kernel descriptor and, per basic block between loop headers, how many VALU / MFMA / LDS /
VMEM / SALU / wait / barrier instructions it holds, plus the issue-cycle estimate of MI355X_MICROARCH.md's constants table
(4 cycles per VALU, 8 per MFMA issue, 8 per transcendental, 32 per MFMA of matrix pipe).
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop") or op.startswith("s_setprio") or op.startswith("s_sleep"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    if dump:
        open(dump, "w").write("\n".join(body))
    for i in range(end, min(end + 400, len(lines))):
        m = re.search(r"\.amdhsa_(next_free_vgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size|next_free_sgpr)\s+(\S+)", lines[i])
        if m:
            print(f"  {m.group(1)} = {m.group(2)}")
        if ".end_amdhsa_kernel" in lines[i]:
            break
    # blocks
    blocks, cur, name = [], Counter(), "entry"
    ops_in = Counter()
    depth_info = {}
    for l in body[1:]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur, ops_in))
            name, cur, ops_in = m.group(1), Counter(), Counter()
            continue
        if "Loop Header" in s or "Parent Loop" in s or "Inner Loop" in s:
            depth_info.setdefault(name, []).append(s.lstrip("; ").split("Depth")[0].strip() + (" Depth" + s.split("Depth")[1] if "Depth" in s else ""))
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        cur[classify(op)] += 1
        ops_in[op] += 1
    blocks.append((name, cur, ops_in))
    tot = Counter()
    print(f"{'block':14s} {'valu':>5s} {'trans':>5s} {'mfma':>5s} {'lds':>5s} {'vmem':>5s} {'salu':>5s} {'wait':>5s} {'barr':>5s} {'issue_cyc':>9s} {'mfma_cyc':>8s}  loop")
    for name, c, ops in blocks:
        n = sum(c.values())
        if n == 0:
            continue
        tot.update(c)
        issue = 4 * c["valu"] + 8 * c["trans"] + 8 * c["mfma"] + 4 * c["lds"] + 4 * c["vmem"]
        if n >= 20 or "--all" in sys.argv:
            print(f"{name:14s} {c['valu']:5d} {c['trans']:5d} {c['mfma']:5d} {c['lds']:5d} {c['vmem']:5d} {c['salu']:5d} {c['wait']:5d} {c['barrier']:5d} {issue:9d} {32 * c['mfma']:8d}  {'; '.join(depth_info.get(name, []))[:60]}")
            if "--ops" in sys.argv and n >= 100:
                print("      " + ", ".join(f"{k} {v}" for k, v in ops.most_common(40)))
    print("total", dict(tot))


if __name__ == "__main__":
    main()
