"""Instruction statistics of one kernel in a device assembly listing (development helper).

    hipcc ... --cuda-device-only -S pt_kernels.hip -o k.s
    python tools/isa_stats.py k.s 'render_fusedILi2ELi0ELi2ELi1E' [--dump out.s]
"""
import collections, re, sys

path, key = sys.argv[1], sys.argv[2]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
lines = open(path).read().split("\n")
start = end = None
for i, l in enumerate(lines):
    if start is None and re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l):
        start = i
    elif start is not None and l.startswith(".Lfunc_end"):
        end = i
        break
body = lines[start:end]
if dump:
    open(dump, "w").write("\n".join(body))
ops = collections.Counter()
cls = collections.Counter()
for l in body:
    m = re.match(r"^\s+([a-z_0-9]+)\s", l + " ")
    if not m or l.lstrip().startswith((".", ";")):
        continue
    op = m.group(1)
    ops[op] += 1
    c = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier")) else
         "smem" if op.startswith(("s_load", "s_buffer")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else
         "branch" if op.startswith(("s_cbranch", "s_branch")) else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cls[c] += 1
print(lines[start].split(":")[0], "lines", len(body))
print(dict(cls), "total", sum(cls.values()))
for k in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32", "v_cndmask_b32_e32", "v_cndmask_b32_e64", "v_mov_b32_e32", "s_mov_b32", "s_mov_b64"):
    print(f"  {k}: {ops[k]}", end="")
print()
# metadata of the kernel (the YAML keys of one kernel are sorted: group_segment... comes before .name, the others after)
name = lines[start].split(":")[0]
for i, l in enumerate(lines):
    if l.strip().startswith(".name:") and l.strip().endswith(name):
        for j in range(i - 1, max(0, i - 40), -1):
            if lines[j].strip().startswith("- .args") or ".name:" in lines[j]:
                break
            if "group_segment_fixed" in lines[j]:
                print("  ", lines[j].strip())
        for j in range(i + 1, min(len(lines), i + 40)):
            if lines[j].strip().startswith("- .args") or ".name:" in lines[j]:
                break
            if any(t in lines[j] for t in ("sgpr_count", "vgpr_count", "spill_count", "private_segment_fixed")):
                print("  ", lines[j].strip())
        break
