#!/usr/bin/env python3
"""Post-compile gate for the kernels that wait for inline-assembly loads with hand-counted vmcnt values -- k_mfma_conv_z
(kernels_mfma_conv.hip) and every instantiation of k_mfma_wgrad_z (kernels_mfma_wgrad_z.hip) -- run by build.sh on the device
assembly of those files:   check_asm_loads.py conv_z <asm> <json>   |   check_asm_loads.py wgrad_z <asm> <json>   |   check_asm_loads.py conv_zdma <asm> <json>
(conv_zdma: k_mfma_conv_z16 / k_mfma_conv_z32 of kernels_mfma_conv_z16.hip, LDS-DMA planes: see scan_dma)

What follows describes the conv_z rules; wgrad_z is checked the same way with its own expectations (per instantiation: the
inline-asm loads come in groups of NL = loads per step, every hand-placed wait is vmcnt(NL) or vmcnt(0), no compiler-generated
vector-memory load and no store between the first inline-asm load and the last hand-placed wait) and with the in-flight rule in
its queue form: a hand-placed `s_waitcnt vmcnt(N)` retires all but the N youngest loads seen so far; a register is "in flight"
from its load to that retirement, and NO instruction other than the loads themselves may read or write it meanwhile.

The kernel issues its plane loads as inline assembly and waits for them with hand-counted `s_waitcnt vmcnt(3/5/7)`: hipcc neither
counts those loads nor protects their destination registers (cdna_hip_programming.md section 5.7).  The counts are only right --
and the registers only safe -- while the compiler adds no vector-memory operation of its own to a step and never touches a load's
destination between the load and the ds_write that consumes it.  Nothing in the language enforces that, so this script checks
the emitted code and fails the build when an assumption no longer holds:

  1. no scratch: .vgpr_spill_count / .sgpr_spill_count / .private_segment_fixed_size are 0 and no scratch_* instruction exists
     (a spill is a vector-memory operation the hand-placed counts do not include);
  2. the hand-written memory operations are all there and nothing was split or duplicated: 12 inline-asm global_load_dwordx4
     (2 prologue + 2 step instantiations x 3 units), 8 buffer_store_dwordx2 (4 plane instantiations x 2 voxels), the six
     hand-placed waits vmcnt(3/5/7) twice each;
  3. register safety, scanned in layout order from the first inline-asm load: a register that is the destination of an
     inline-asm load is "in flight" until a ds_write_b128 takes it as its data operand; no other instruction may read or write
     it while in flight (a v_mov copy, a spill, or reuse as a temporary would read or clobber data that has not landed);
  4. the toolchain that produced this result is recorded next to the library (conv_z_check.json).

UNET_NO_CONV_Z=1 (environment, read by the engine) is the documented fallback: the halo-tile kernel k_mfma_conv_p."""
import json
import re
import subprocess
import sys



def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def kernel_bodies(text, match):
    """[(symbol, lines)] of every kernel whose mangled name contains `match`"""
    out = []
    for m in re.finditer(r"\n(_Z[A-Za-z0-9_]*%s[A-Za-z0-9_]*):" % re.escape(match), text):
        start = m.start()
        end = text.find("s_endpgm", start)
        out.append((m.group(1), text[start:end].split("\n")))
    return out


def metadata(text, sym):
    meta = {}
    for blk in text.split("  - .agpr_count:")[1:]:
        if ("%s.kd" % sym) in blk:
            for key in (".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".vgpr_count"):
                m = re.search(r"%s:\s+(\d+)" % re.escape(key), blk)
                meta[key] = int(m.group(1)) if m else None
            break
    return meta


def scan(body, mode):
    """mode 'consume' (conv_z): a load's registers are in flight until a ds_write_b128 takes them as data.
    mode 'retire' (wgrad_z): hand-placed vmcnt(N) retires all but the N youngest loads; ds_write may only read retired registers."""
    errors, in_asm = [], False
    asm_loads = stores = other_vmem = 0
    waits = {}
    queue = []          # retire mode: destination register sets in issue order
    inflight = set()
    seen_first = False
    last_wait_line = max([i for i, l in enumerate(body) if "s_waitcnt vmcnt" in l] or [0])
    for ln, line in enumerate(body):
        t = line.strip()
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        op = t.split()[0]
        args = t[len(op):]
        if op.startswith("scratch_"):
            errors.append("scratch instruction: %s" % t)
        if op.startswith("buffer_store") or op.startswith("global_store"):
            if op == "buffer_store_dwordx2":
                stores += 1
            if seen_first and ln < last_wait_line and (mode == "retire" or (op.startswith("buffer_store") and op != "buffer_store_dwordx2")):
                errors.append("line %d: unexpected store between the asm loads and their waits: %s" % (ln, t))
        if in_asm and op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                waits[n] = waits.get(n, 0) + 1
                if mode == "retire":
                    keep = queue[len(queue) - n:] if n else []
                    queue = keep
                    inflight = set().union(*keep) if keep else set()
            continue
        if in_asm and op == "global_load_dwordx4":
            asm_loads += 1
            seen_first = True
            parts = args.split(",")
            dst, addr = regs_of(parts[0]), regs_of(",".join(parts[1:]))
            hit = (dst | addr) & inflight
            if hit:
                errors.append("line %d: load touches registers still in flight %s: %s" % (ln, sorted(hit), t))
            inflight |= dst
            queue.append(dst)
            continue
        if not in_asm and seen_first and ln < last_wait_line and (op.startswith("global_load") or op.startswith("buffer_load")) and mode == "retire":
            other_vmem += 1
            errors.append("line %d: compiler-generated vector load inside the hand-counted region: %s" % (ln, t))
        used = regs_of(args)
        if op == "ds_write_b128" and mode == "consume":
            parts = args.split(",")
            data = regs_of(parts[1]) if len(parts) > 1 else set()
            if regs_of(parts[0]) & inflight:
                errors.append("line %d: ds_write address register in flight: %s" % (ln, t))
            inflight -= data
            continue
        hit = used & inflight
        if hit:
            errors.append("line %d: %s touches load destinations in flight %s" % (ln, t, sorted(hit)))
    return errors, {"asm_loads": asm_loads, "buffer_store_dwordx2": stores, "hand_waits": waits}


def scan_dma(body):
    """k_mfma_conv_z16 / _z32: planes travel by inline-asm LDS-DMA (global_load_lds_dwordx4, no destination registers) and are awaited
    with ONE hand-counted value vmcnt(N), N = (P + S) * PF - P for P DMA pieces and S output stores per step, PF planes ahead (younger than a
    plane's last piece: that step's S stores and PF - 1 whole steps).
    Checked between consecutive hand-placed waits vmcnt(N): exactly P LDS-DMA pieces and S buffer_store_dwordx2, and no other
    vector-memory instruction except the buffer_load_dwordx2 of the accumulate path (an extra operation there only makes the
    count conservative, and the compiler follows it with its own vmcnt(0))."""
    errors, in_asm = [], False
    waits, seg, segs = {}, None, []
    for line in body:
        t = line.strip()
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        op = t.split()[0]
        if in_asm and op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                waits[n] = waits.get(n, 0) + 1
                if seg is not None:
                    segs.append(seg)
                seg = {"dma": 0, "store": 0, "other": []} if n else None
            continue
        if seg is None:
            continue
        if seg.get("closed"):
            continue
        if (op.startswith("s_cbranch_scc") or op == "s_branch") and seg["store"] >= 2:
            seg["closed"] = True       # the step's straight-line code ends at the loop-control branch that follows its stores; what the
            continue                   # layout places behind it (the next item's set-up) runs after the loop and ends in a vmcnt(0)
        if op.startswith("global_load_lds"):
            seg["dma"] += 1
        elif op == "buffer_store_dwordx2":
            seg["store"] += 1
        elif op.startswith("scratch_") or ((op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_")) and op != "buffer_load_dwordx2"):
            seg["other"].append(t)
    steady = [n for n in waits if n]
    st = {"hand_waits": waits, "steps_checked": len(segs)}
    if len(steady) != 1 or not waits.get(0):
        errors.append("hand-placed waits %s (expected one steady-state value and vmcnt(0))" % waits)
        return errors, st
    n = steady[0]
    shapes = set((g["dma"], g["store"]) for g in segs)
    for g in segs:
        if g["other"]:
            errors.append("vector-memory instruction inside a hand-counted step: %s" % g["other"][0])
            break
    if len(shapes) != 1:
        errors.append("steps differ in their (DMA pieces, stores): %s" % sorted(shapes))
    else:
        pcs, sts = next(iter(shapes))
        st.update({"dma_pieces_per_step": pcs, "stores_per_step": sts})
        pf = (n + pcs) / float(pcs + sts) if pcs + sts else 0
        if pcs == 0 or pf != int(pf) or pf < 1:
            errors.append("vmcnt(%d) is not (P + S) * PF - P for P = %d pieces, S = %d stores" % (n, pcs, sts))
        else:
            st["planes_ahead"] = int(pf)
    return errors, st


def scan_dma_cfg(body):
    """kernels_mfma_s2.hip (k_s2_scatter / k_s2_gather): the LDS-DMA discipline of scan_dma, checked over the control-flow graph
    instead of the layout order (the compiler rotates the unrolled step loop and sinks a step's last store below the loop-exit
    test, so straight-line counting between two waits no longer sees whole steps).
    From every hand-placed wait all paths are followed to the next hand-placed wait: an exec-mask branch is followed on the side
    where lanes are active (every wave issues every DMA piece: static_assert in the kernel), a scalar conditional branch on both
    sides, s_branch to its target.  Required: every path from a steady wait vmcnt(N) to a steady wait carries exactly P LDS-DMA
    pieces and S buffer_store_dwordx2 and no other vector-memory instruction, N = (P + S) PF - P for an integer PF >= 1, and every
    path from a vmcnt(0) to the first steady wait (the prologue) carries PF P pieces and PF S stores."""
    errors = []
    ins, labels, in_asm = [], {}, False
    for line in body:
        t = line.strip()
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not t or t[0] in ";":
            continue
        if t.endswith(":") or re.match(r"^\.?[A-Za-z_][\w.$]*:", t):
            labels[t.split(":")[0]] = len(ins)
            continue
        if t[0] == ".":
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        ins.append((t.split()[0], t, in_asm))
    waits = {}
    wait_at = {}
    for i, (op, t, a) in enumerate(ins):
        if a and op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                wait_at[i] = int(m.group(1))
                waits[wait_at[i]] = waits.get(wait_at[i], 0) + 1
    steady = sorted(n for n in waits if n)
    st = {"hand_waits": waits}
    if len(steady) != 1 or not waits.get(0):
        errors.append("hand-placed waits %s (expected one steady-state value and vmcnt(0))" % waits)
        return errors, st
    N = steady[0]
    results = {}      # (from kind, to kind) -> set of (dma, store)
    other = []
    sys.setrecursionlimit(100000)
    starts = list(wait_at.items()) + [(-1, 0)]      # the kernel's entry counts as a vmcnt(0): nothing is in flight there
    for start, n0 in starts:
        seen = set()
        stack = [(start + 1, 0, 0)]
        while stack:
            i, d, sto = stack.pop()
            while True:
                if i >= len(ins):
                    break
                if (i, d, sto) in seen:
                    break
                seen.add((i, d, sto))
                op, t, a = ins[i]
                if i in wait_at:
                    results.setdefault((n0, wait_at[i]), set()).add((d, sto))
                    break
                if op == "s_endpgm":
                    break
                if op.startswith("global_load_lds"):
                    d += 1
                elif op == "buffer_store_dwordx2":
                    sto += 1
                elif op.startswith("scratch_") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
                    if n0 == N:
                        other.append(t)
                if op == "s_branch":
                    i = labels.get(t.split()[1], len(ins))
                    continue
                if op in ("s_cbranch_execz", "s_cbranch_execnz"):
                    i = labels.get(t.split()[1], len(ins)) if op == "s_cbranch_execnz" else i + 1
                    continue
                if op.startswith("s_cbranch"):
                    stack.append((labels.get(t.split()[1], len(ins)), d, sto))
                i += 1
    if other:
        errors.append("vector-memory instruction inside a hand-counted step: %s" % other[0])
    shapes = results.get((N, N), set())
    st["steps_checked"] = sum(1 for k in wait_at.values() if k == N)
    if len(shapes) != 1:
        errors.append("paths between two steady waits differ in their (DMA pieces, stores): %s" % sorted(shapes))
        return errors, st
    pcs, sts = next(iter(shapes))
    st.update({"dma_pieces_per_step": pcs, "stores_per_step": sts})
    pf = (N + pcs) / float(pcs + sts) if pcs + sts else 0
    if pcs == 0 or pf != int(pf) or pf < 1:
        errors.append("vmcnt(%d) is not (P + S) * PF - P for P = %d pieces, S = %d stores" % (N, pcs, sts))
        return errors, st
    pf = int(pf)
    st["planes_ahead"] = pf
    pro = results.get((0, N), set())
    if pro != {(pf * pcs, pf * sts)}:
        errors.append("prologue paths carry %s (expected %s)" % (sorted(pro), (pf * pcs, pf * sts)))
    for (a0, b0), sh in results.items():      # a step that leaves the loop may do so only after its full complement (it ends in vmcnt(0))
        if a0 == N and b0 == 0 and any(x != (pcs, sts) for x in sh):
            errors.append("a step's path to the final vmcnt(0) carries %s" % sorted(sh))
    return errors, st


def main(which, path, out_json=None):
    text = open(path).read()
    ver = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout.strip().split("\n")
    recs, failed = [], False
    if which == "conv_z":
        bodies = kernel_bodies(text, "k_mfma_conv_zE")
        mode = "consume"
    elif which == "conv_zdma":
        bodies = kernel_bodies(text, "k_mfma_conv_z16") + kernel_bodies(text, "k_mfma_conv_z32")
        mode = "dma"
    elif which == "wgrad_zd":
        # kernels_mfma_wgrad_zd.hip: P DMA pieces per wave and step, no stores in the loop, one hand-counted wait vmcnt((PF - 1) P)
        bodies = kernel_bodies(text, "k_mfma_wgrad_zdI")
        mode = "dma"
    elif which == "s2dma":
        # kernels_mfma_s2.hip: the same LDS-DMA discipline (P pieces + S stores per wave and step, one hand-counted wait)
        bodies = kernel_bodies(text, "k_s2_scatter") + kernel_bodies(text, "k_s2_gather")
        mode = "dma"
    else:
        bodies = kernel_bodies(text, "k_mfma_wgrad_zI")
        mode = "retire"
    if not bodies:
        sys.exit("check_asm_loads: no %s kernel found in %s" % (which, path))
    for sym, body in bodies:
        meta = metadata(text, sym)
        if mode == "dma":
            errors, st = scan_dma_cfg(body) if which in ("s2dma", "wgrad_zd") else scan_dma(body)     # (spills outside the hand-counted steps are harmless here: the steps are checked instruction by instruction)
            st.update({"asm_loads": st.get("dma_pieces_per_step", 0), "buffer_store_dwordx2": st.get("stores_per_step", 0)})
        else:
            errors, st = scan(body, mode)
            for key in (".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size"):
                if meta.get(key) != 0:
                    errors.append("%s = %s (must be 0)" % (key, meta.get(key)))
        if mode == "dma":
            if which in ("s2dma", "wgrad_zd") and (meta.get(".vgpr_spill_count") or meta.get(".private_segment_fixed_size")):
                errors.append("scratch in use: vgpr spills %s, private segment %s" % (meta.get(".vgpr_spill_count"), meta.get(".private_segment_fixed_size")))
        elif which == "conv_z":
            if st["asm_loads"] != 12:
                errors.append("inline-asm plane loads: %d (expected 12)" % st["asm_loads"])
            if st["buffer_store_dwordx2"] != 8:
                errors.append("buffer_store_dwordx2: %d (expected 8)" % st["buffer_store_dwordx2"])
            if any(st["hand_waits"].get(k, 0) != 2 for k in (3, 5, 7)):
                errors.append("hand-placed waits vmcnt(3/5/7): %s (expected 2 each)" % st["hand_waits"])
        else:
            # PD = 2 sets: 8 fetch sites (2 prologue + 6 unrolled steps); PD = 3: 6 sites (3 + 3).  Every hand-placed wait is
            # vmcnt(0), vmcnt(NL) or vmcnt(2 NL) with NL = loads per fetch.
            import math
            w = st["hand_waits"]
            nz = [k for k in w if k]
            nl = 0
            for k in nz:
                nl = math.gcd(nl, k)
            if not nl or st["asm_loads"] % nl or st["asm_loads"] // nl not in (6, 8):
                errors.append("inline-asm loads: %d with waits %s (expected 6 or 8 fetch sites x NL)" % (st["asm_loads"], w))
            elif set(w) - {0, nl, 2 * nl} or not w.get(0) or not w.get(nl):
                errors.append("hand-placed waits: %s (expected only vmcnt(0), vmcnt(%d), vmcnt(%d))" % (w, nl, 2 * nl))
        rec = dict(kernel=sym, ok=not errors, errors=errors, vgpr_count=meta.get(".vgpr_count"), **st)
        recs.append(rec)
        failed = failed or bool(errors)
    out = {"checked": which, "ok": not failed, "kernels": recs, "validated_with": ver[:2]}
    if out_json:
        json.dump(out, open(out_json, "w"), indent=1)
    if failed:
        print("check_asm_loads: %s no longer satisfies the assumptions of its hand-placed waits:" % which, file=sys.stderr)
        for r in recs:
            for e in r["errors"][:12]:
                print("   %s: %s" % (r["kernel"][-40:], e), file=sys.stderr)
        print("   (UNET_NO_CONV_Z=1 / UNET_NO_WGRAD_Z=1 select the halo-tile kernels instead; fix the kernel or the counts before shipping)", file=sys.stderr)
        sys.exit(1)
    print("check_asm_loads: %s ok (%s; %s)" % (which, ", ".join("%s VGPRs / %d asm loads" % (r["vgpr_count"], r["asm_loads"]) for r in recs), ver[0] if ver else "?"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
