#!/usr/bin/env python3
"""Post-compile gate for k_mfma_conv_z (kernels_mfma_conv.hip), run by build.sh on the device assembly of that file.

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

KERNEL = "_ZN4unet13k_mfma_conv_zENS_12MfmaConvArgsENS_5ZWorkE"


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def main(path, out_json=None):
    text = open(path).read()
    start = text.find("\n" + KERNEL + ":")
    if start < 0:
        sys.exit("check_conv_z: kernel symbol not found in %s" % path)
    end = text.find("s_endpgm", start)
    body = text[start:end].split("\n")
    errors = []
    # 1. metadata
    meta = {}
    for key in (".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".vgpr_count"):
        m = re.search(r"\.name:\s+%s\b.*?%s:\s+(\d+)" % (re.escape(KERNEL), re.escape(key)), text, re.S)
        m2 = None
        if not m:   # the metadata block lists keys alphabetically: search the block that names this kernel
            for blk in text.split("  - .agpr_count:")[1:]:
                if KERNEL in blk.split(".symbol:")[0] or ("%s.kd" % KERNEL) in blk:
                    m2 = re.search(r"%s:\s+(\d+)" % re.escape(key), blk)
                    break
        val = int(m.group(1)) if m else (int(m2.group(1)) if m2 else None)
        meta[key] = val
    for key in (".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size"):
        if meta.get(key) not in (0,):
            errors.append("%s = %s (must be 0)" % (key, meta.get(key)))
    # 2./3. instruction scan
    in_asm = False
    asm_loads = stores = 0
    waits = {3: 0, 5: 0, 7: 0}
    inflight = set()
    seen_first = False
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
        if op.startswith("scratch_"):
            errors.append("scratch instruction: %s" % t)
        if op.startswith("buffer_store") or op.startswith("global_store"):
            if op == "buffer_store_dwordx2":
                stores += 1
            elif seen_first and op.startswith("buffer_store"):
                errors.append("unexpected store form in the plane loop: %s" % t)
        if in_asm and op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m and int(m.group(1)) in waits:
                waits[int(m.group(1))] += 1
        args = t[len(op):]
        if in_asm and op == "global_load_dwordx4":
            asm_loads += 1
            seen_first = True
            dst = regs_of(args.split(",")[0])
            addr = regs_of(",".join(args.split(",")[1:]))
            hit = (dst | addr) & inflight
            if hit:
                errors.append("line %d: load touches registers still in flight %s: %s" % (ln, sorted(hit), t))
            inflight |= dst
            continue
        used = regs_of(args)
        if op == "ds_write_b128":
            parts = args.split(",")
            data = regs_of(parts[1]) if len(parts) > 1 else set()
            addr = regs_of(parts[0])
            if addr & inflight:
                errors.append("line %d: ds_write address register in flight: %s" % (ln, t))
            inflight -= data
            continue
        hit = used & inflight
        if hit:
            errors.append("line %d: %s touches load destinations in flight %s" % (ln, t, sorted(hit)))
    if asm_loads != 12:
        errors.append("inline-asm plane loads: %d (expected 12)" % asm_loads)
    if stores != 8:
        errors.append("buffer_store_dwordx2: %d (expected 8)" % stores)
    if any(v != 2 for v in waits.values()):
        errors.append("hand-placed waits vmcnt(3/5/7): %s (expected 2 each)" % waits)
    ver = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout.strip().split("\n")
    rec = {"kernel": "k_mfma_conv_z", "ok": not errors, "errors": errors, "vgpr_count": meta.get(".vgpr_count"),
           "asm_plane_loads": asm_loads, "buffer_store_dwordx2": stores, "hand_waits": waits, "validated_with": ver[:2]}
    if out_json:
        json.dump(rec, open(out_json, "w"), indent=1)
    if errors:
        print("check_conv_z: k_mfma_conv_z no longer satisfies the assumptions of its hand-placed waits:", file=sys.stderr)
        for e in errors[:20]:
            print("   " + e, file=sys.stderr)
        print("   (UNET_NO_CONV_Z=1 selects the halo-tile kernel instead; fix the kernel or the counts before shipping)", file=sys.stderr)
        sys.exit(1)
    print("check_conv_z: ok (%s VGPRs, 12 asm loads, 8 stores; %s)" % (meta.get(".vgpr_count"), ver[0] if ver else "?"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
