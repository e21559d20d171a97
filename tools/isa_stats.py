#!/usr/bin/env python3
"""Static ISA summary of the product kernels (cross-compiled, no GPU): registers, spills and the
per-basic-block instruction mix of one kernel.  usage: tools/isa_stats.py [kernel-name-substring]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "build", "asm"); os.makedirs(out, exist_ok=True)
asm = os.path.join(out, "accel.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC",
                       "-Wno-unused-function", "--cuda-device-only", "-S", "-o", asm, "sgtd_accel.hip"],
                      cwd=os.path.join(root, "sgtd_amd", "csrc"), stderr=subprocess.DEVNULL)
s = open(asm).read()
want = sys.argv[1] if len(sys.argv) > 1 else None
# function bodies: "<name>:" ... ".Lfunc_endN:"; resource blocks: ".amdhsa_kernel <name>" ... ".end_amdhsa_kernel"
bodies, metas = {}, {}
lines = s.split('\n')
cur = None
for l in lines:
    m = re.match(r'^(_Z\S+):', l)
    if m: cur = m.group(1); bodies[cur] = []; continue
    if l.startswith('.Lfunc_end'): cur = None
    if cur: bodies[cur].append(l)
cur = None
for l in lines:
    m = re.match(r'^\s*\.amdhsa_kernel (\S+)', l)
    if m: cur = m.group(1); metas[cur] = []; continue
    if '.end_amdhsa_kernel' in l: cur = None
    if cur: metas[cur].append(l)
for name in bodies:
    if name not in metas: continue
    body, meta = '\n'.join(bodies[name]), '\n'.join(metas[name])
    v = (re.search(r"; NumVgprs: (\d+)", s[s.index(name + ":"):]) or re.search(r"next_free_vgpr (\d+)", meta)).group(1); sg = re.search(r'next_free_sgpr (\d+)', meta).group(1)
    lanes = len(re.findall(r'v_(read|write)lane', body)); scr = len(re.findall(r'scratch_', body))
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split('(')[0]
    print("%-60s vgpr %3s sgpr %3s valu %5d salu %5d lane-ops %4d scratch %d" % (short[:60], v, sg, len(re.findall(r'^\s*v_', body, re.M)), len(re.findall(r'^\s*s_', body, re.M)), lanes, scr))
    if want and want in short:
        cur = ["entry", []]; blocks = []
        for l in body.split('\n'):
            mm = re.match(r'^(\.LBB\d+_\d+):', l)
            if mm: blocks.append(cur); cur = [mm.group(1), []]
            else: cur[1].append(l)
        blocks.append(cur)
        for bn, ls in blocks:
            nv = sum(1 for l in ls if re.match(r'\s*v_', l)); ns = sum(1 for l in ls if re.match(r'\s*s_', l))
            if nv + ns < 12: continue
            print("   %-12s valu %3d salu %3d lane-ops %3d vmem-ld %d vmem-st %d lds %d" % (bn, nv, ns, sum(1 for l in ls if re.search(r'v_(read|write)lane', l)),
                  sum(1 for l in ls if 'global_load' in l), sum(1 for l in ls if 'global_store' in l or 'global_atomic' in l), sum(1 for l in ls if re.match(r'\s*ds_', l))))
