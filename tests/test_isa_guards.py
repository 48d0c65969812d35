"""ISA guards (CPU side: hipcc cross-compiles gfx950 without a GPU).  The evaluation kernels are bit-exact only while a few
properties of their machine code hold that no runtime test pins down directly; a compiler update that breaks one of them should
fail HERE, not as a corrupted score somewhere:

  * no scratch and no spills (a spilled tap or window register costs the fade loop 2-3x, and the ablation notes in profiles/ assume none),
  * VGPRs within the occupancy each kernel is designed for,
  * no MFMA (these are per-pixel private 25-tap kernels in a prescribed summation order),
  * the linear kernel's raw-sample loads are inline assembly with a hand-placed vmcnt wait (the compiler neither counts nor waits for
    them): no instruction may read a register such a load writes before the next vmcnt wait -- a copy of a loaded value placed behind
    the load by the register allocator would read stale data (seen while the loop was written),
  * and the one that can corrupt silently: eval_tile_stage.h places `s_waitcnt lgkmcnt(N)` with N > 0 by hand to consume LDS reads in
    issue order while later ones are still in flight.  LDS operations of a wave return in order, scalar memory loads do NOT, and both
    count in lgkmcnt -- a partial wait is only meaningful while no s_load / s_buffer_load is outstanding.  The check walks the
    control-flow graph of every kernel: "a scalar load may be outstanding" is propagated from block to block to a fixed point, and no
    partial lgkmcnt wait may be reached in that state.
"""
import hashlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "amatsukaze_amd", "csrc")
CACHE = os.path.join("/tmp", "amt_isa_cache")

# file -> (extra flags, {kernel-name substring: max VGPRs})   budgets: 512 VGPRs per SIMD lane / waves per SIMD, rounded to the allocation granule of 8
FILES = {
    "eval_linear_kernels.hip": (["-fno-slp-vectorize"], {"logo_eval_linear_kernel16": 128, "logo_eval_linear_kernel": 128}),
    "eval_pair_kernels.hip": ([], {"logo_eval_pair_kernel": 168}),
    "eval_fused_kernels.hip": (["-mllvm", "-amdgpu-sched-strategy=max-ilp"], {"logo_eval_fused_kernel": 256}),
    "stats_kernels.hip": ([], {"frame_stats_kernel": 256}),
    "erase_scan_kernels.hip": ([], {"delogo_kernel": 128, "calc_fades_kernel": 128}),
}


def compile_asm(name):
    from amatsukaze_amd import build as B
    src = os.path.join(CSRC, name)
    flags = [f for f in B.FLAGS if f not in ("-fPIC",)] + FILES[name][0]
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    h = hashlib.sha256(" ".join(flags).encode())
    for d in sorted(deps):
        h.update(open(d, "rb").read())
    os.makedirs(CACHE, exist_ok=True)
    out = os.path.join(CACHE, f"{name}.{h.hexdigest()[:16]}.s")
    if not os.path.exists(out):
        subprocess.check_call([B.hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels_of(asm):
    """{mangled kernel name: {'body': [instruction / label lines], 'meta': {...}}}"""
    meta = {}
    for m in re.finditer(r"- \.agpr_count:.*?\n(?=  - \.agpr_count:|\.\.\.|amdhsa\.target)", asm, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1)) for k in
                      ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "agpr_count")}
    out = {}
    for name in meta:
        m = re.search(rf"^{re.escape(name)}:[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M)
        assert m, f"no body for {name}"
        body = [l.split(";")[0].rstrip() for l in m.group(1).splitlines()]
        out[name] = {"meta": meta[name], "body": [l for l in body if l.strip()]}
    return out


def partial_waits_with_smem_outstanding(body):
    """CFG walk: returns the partial lgkmcnt waits that can be reached with a scalar load outstanding"""
    # basic blocks: split at labels and after branches
    blocks, cur, label_of = [], {"label": None, "ins": []}, {}
    for l in body:
        t = l.strip()
        if re.match(r"^\.?[A-Za-z_][\w$.]*:$", t):
            if cur["ins"] or cur["label"] is not None:
                blocks.append(cur)
            cur = {"label": t[:-1], "ins": []}
            continue
        if t.startswith("."):            # directives
            continue
        cur["ins"].append(t)
        if re.match(r"^s_(c?branch|endpgm|setpc|swappc)", t):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"] or cur["label"] is not None:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        last = b["ins"][-1] if b["ins"] else ""
        s = []
        m = re.match(r"^s_(branch|cbranch_\w+)\s+(\S+)", last)
        if m:
            assert m.group(2) in label_of, f"branch to unknown label {m.group(2)}"
            s.append(label_of[m.group(2)])
            if m.group(1) != "branch" and i + 1 < len(blocks):
                s.append(i + 1)
        elif re.match(r"^s_(endpgm|setpc|swappc)", last):
            pass
        elif i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)

    def transfer(state, ins, hits=None):
        for t in ins:
            if re.match(r"^s_(load|buffer_load|scratch_load|atomic|buffer_atomic|dcache|memtime|memrealtime|getreg_b32 .*HW_REG_SHADER_CYCLES)", t):
                state = True
            m = re.match(r"^s_waitcnt\b(.*)", t)
            if m:
                lg = re.search(r"lgkmcnt\((\d+)\)", m.group(1))
                if lg and int(lg.group(1)) == 0:
                    state = False
                elif lg and state and hits is not None:
                    hits.append(t)
                elif not lg and re.search(r"^\s*(0x[0-9a-fA-F]+|\d+)\s*$", m.group(1)):      # raw immediate: decode lgkmcnt (bits 11:8)
                    v = int(m.group(1).strip(), 0)
                    n = (v >> 8) & 0xF
                    if n == 0:
                        state = False
                    elif n < 15 and state and hits is not None:
                        hits.append(t)
        return state

    entry = [False] * len(blocks)
    changed = True
    while changed:
        changed = False
        for i, b in enumerate(blocks):
            o = transfer(entry[i], b["ins"])
            for j in succ[i]:
                if o and not entry[j]:
                    entry[j] = True
                    changed = True
    hits = []
    for i, b in enumerate(blocks):
        transfer(entry[i], b["ins"], hits)
    return hits, len(blocks)


def reads_of_registers_in_flight(raw_body):
    """raw_body: the kernel's lines WITH comments (;;#ASMSTART / ;;#ASMEND mark inline assembly).  Walks the text in order: a buffer_ /
    global_ load inside an inline-assembly block puts its destination registers "in flight" until the next `s_waitcnt ... vmcnt`; any
    instruction that names one of them as a source before that is returned.  (Linear scan: the loop's loads are issued at its end and
    waited for one trip later, so the scan wraps round once.)"""
    def regs(tok):
        out = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", tok):
            out |= set(range(int(a), int(b) + 1))
        for a in re.findall(r"\bv(\d+)\b", tok):
            out.add(int(a))
        return out
    hits, pending, inasm = [], {}, False
    for l in raw_body:
        if "#ASMSTART" in l:
            inasm = True
            continue
        if "#ASMEND" in l:
            inasm = False
            continue
        t = l.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        op = t.split()[0]
        if op == "s_waitcnt" and "vmcnt" in t:
            pending.clear()
            continue
        ops = t[len(op):].split(",")
        if inasm and re.match(r"(buffer_load|global_load)", op):
            srcs = ops[1:]
            for o in srcs:
                for r in regs(o):
                    if r in pending:
                        hits.append((t, f"v{r}", pending[r]))
            for r in regs(ops[0]):
                pending[r] = t
            continue
        nodst = op.startswith(("ds_write", "buffer_store", "global_store", "scratch_store", "s_", "v_cmp", "v_cmpx"))
        for o in (ops if nodst else ops[1:]):
            for r in regs(o):
                if r in pending:
                    hits.append((t, f"v{r}", pending[r]))
    return hits


def raw_kernel_bodies(asm):
    return {m.group(1): m.group(2).splitlines() for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M)}


def test_in_flight_scan_sees_a_planted_copy():
    good = [";;#ASMSTART", "buffer_load_dword v5, v1, s[0:3], 0 offen", ";;#ASMEND", "v_add_f32 v2, v3, v4", ";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND",
            "v_mov_b32 v6, v5"]
    bad = [";;#ASMSTART", "buffer_load_dword v5, v1, s[0:3], 0 offen", ";;#ASMEND", "v_mov_b32 v6, v5", ";;#ASMSTART", "s_waitcnt vmcnt(0)", ";;#ASMEND"]
    assert not reads_of_registers_in_flight(good) and reads_of_registers_in_flight(bad)
    # a compiler load is not tracked (the compiler waits for its own loads)
    assert not reads_of_registers_in_flight(["global_load_dword v5, v1, s[0:1]", "s_waitcnt vmcnt(0)", "v_mov_b32 v6, v5"])


def test_linear_kernel_reads_no_register_in_flight():
    asm = compile_asm("eval_linear_kernels.hip")
    bodies = {k: v for k, v in raw_kernel_bodies(asm).items() if "logo_eval_linear_kernel" in k}
    assert len(bodies) == 2
    for kname, body in bodies.items():
        nasm = sum(1 for i, l in enumerate(body) if re.match(r"\s*buffer_load", l) and i > 0 and "#ASMSTART" in body[i - 1])
        assert nasm >= 8, f"{kname}: the raw-sample loads are no longer inline assembly ({nasm})"
        hits = reads_of_registers_in_flight(body)
        assert not hits, f"{kname}: a register is read while its load is in flight: {hits[:3]}"


def scratch_inside_loops(body):
    """scratch instructions between a label and a later branch back to it"""
    lines = [l.strip() for l in body]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    hits = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            hits += [x for x in lines[labels[m.group(1)]:i] if "scratch_" in x]
    return hits


def test_scratch_loop_scan_sees_a_planted_reload():
    assert scratch_inside_loops([".LBB0_1:", "scratch_load_dword v1, off, off", "s_cbranch_scc1 .LBB0_1"])
    assert not scratch_inside_loops(["scratch_store_dword off, v1, off", ".LBB0_1:", "v_add_f32 v0, v0, v1", "s_cbranch_scc1 .LBB0_1", "scratch_load_dword v1, off, off"])


def test_cfg_walker_sees_a_planted_violation():
    bad = ["s_load_dword s0, s[2:3], 0x0", "ds_read_b64 v[0:1], v2", "s_waitcnt lgkmcnt(1)", "s_endpgm"]
    ok = ["s_load_dword s0, s[2:3], 0x0", "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[0:1], v2", "ds_read_b64 v[2:3], v2", "s_waitcnt lgkmcnt(1)", "s_endpgm"]
    loop = [".LBB0_1:", "ds_read_b64 v[0:1], v2", "ds_read_b64 v[2:3], v2", "s_waitcnt lgkmcnt(1)", "s_load_dword s0, s[2:3], 0x0",
            "s_cbranch_scc1 .LBB0_1", "s_endpgm"]          # the load of iteration i is outstanding at the partial wait of iteration i + 1
    assert partial_waits_with_smem_outstanding(bad)[0] and not partial_waits_with_smem_outstanding(ok)[0]
    assert partial_waits_with_smem_outstanding(loop)[0]


@pytest.mark.parametrize("name", sorted(FILES))
def test_kernel_isa_properties(name):
    asm = compile_asm(name)
    ks = kernels_of(asm)
    budgets = FILES[name][1]
    seen = set()
    for kname, k in ks.items():
        m = k["meta"]
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, f"{kname}: scratch / spills {m}"
        assert not scratch_inside_loops(k["body"]), f"{kname}: scratch access inside a loop"
        # (delogo_kernel, the staging / prologue blocks of the generic fused kernel and the linear kernel -- whose loop carries more
        # wave-uniform state than there are scalar registers -- park scalars in VGPR lanes: v_writelane, no memory.  The pair kernel and
        # the frame metrics must have none)
        if name not in ("erase_scan_kernels.hip", "eval_fused_kernels.hip", "eval_linear_kernels.hip"):
            assert m["sgpr_spill_count"] == 0, f"{kname}: scalar spills {m}"
        assert not any(re.match(r"^\s*v_(mfma|smfmac)", l) for l in k["body"]), f"{kname}: MFMA in a kernel that must not have any"
        assert not any("scratch_" in l for l in k["body"]), f"{kname}: scratch instructions"
        for sub in sorted(budgets, key=len, reverse=True):          # the longest matching name decides (kernel16 before kernel)
            if sub in kname:
                assert m["vgpr_count"] + m["agpr_count"] <= budgets[sub], f"{kname}: {m['vgpr_count']} VGPRs (+{m['agpr_count']} AGPRs) > {budgets[sub]}"
                seen.add(sub)
                break
        if name == "stats_kernels.hip":
            # the row sets must live in registers: a dynamically indexed member once made the compiler park them in LDS (2x slower)
            assert not any(re.match(r"^\s*ds_(read|write)", l) for l in k["body"]), f"{kname}: LDS traffic in the frame metrics"
        hits, nblocks = partial_waits_with_smem_outstanding(k["body"])
        assert nblocks > 0
        assert not hits, f"{kname}: partial lgkmcnt wait with a scalar load possibly outstanding: {hits[:3]}"
    assert seen == set(budgets), f"kernels not found in {name}: {set(budgets) - seen}"
