#!/usr/bin/env python3
"""Build-time audit of the kernels whose inline asm owns accumulator registers: fa_fwd_w64_kernel, its masked
forms (causal, ragged, both) and its key-split form, and the fa_fwd_hd_kernel family.

These kernels name a[..] literally in their instruction strings; the compiler does not know they hold
live values.  That is only sound while the compiler itself uses no accumulator register and spills
nothing in that kernel.  This script compiles the library's device code to assembly and fails if
any of them has scratch, spills, or any v_accvgpr_* / a[..] operand outside the #ASMSTART/#ASMEND
blocks, anywhere in the function (a kernel with early exits has several s_endpgm: the scan runs to the
function's end label).

    python tools/check_owned_regs.py [extra hipcc flags...]

Environment (experiment builds): FA_AUDIT_SRC = another source file, FA_AUDIT_KERNELS = comma-separated mangled
names to audit instead of the library's list, FA_AUDIT_KEEP = where to keep the assembly.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "flash-attention_amd", "csrc", "flash_attention_capi.hip")
KERNELS = ["_ZN2fa17fa_fwd_w64_kernelENS_9FwdParamsE", "_ZN2fa24fa_fwd_w64_causal_kernelENS_9FwdParamsE",
           "_ZN2fa24fa_fwd_w64_ragged_kernelENS_9FwdParamsE", "_ZN2fa31fa_fwd_w64_ragged_causal_kernelENS_9FwdParamsE",
           "_ZN2fa23fa_fwd_w64_split_kernelENS_9FwdParamsE", "_ZN2fa24fa_fwd_w64_varlen_kernelENS_9FwdParamsE", "_ZN2fa31fa_fwd_w64_varlen_causal_kernelENS_9FwdParamsE",
           "_ZN2fa16fa_fwd_hd_kernelILi256EEEvNS_9FwdParamsE",
           "_ZN2fa16fa_fwd_hd_kernelILi64EEEvNS_9FwdParamsE", "_ZN2fa23fa_fwd_hd_masked_kernelILi256EEEvNS_9FwdParamsE",
           "_ZN2fa23fa_fwd_hd_masked_kernelILi64EEEvNS_9FwdParamsE", "_ZN2fa23fa_fwd_hd_masked_kernelILi128EEEvNS_9FwdParamsE",
           "_ZN2fa22fa_fwd_hd_state_kernelILi64EEEvNS_9FwdParamsE", "_ZN2fa22fa_fwd_hd_state_kernelILi128EEEvNS_9FwdParamsE",
           "_ZN2fa22fa_fwd_hd_state_kernelILi256EEEvNS_9FwdParamsE"]


def main():
    src = os.environ.get("FA_AUDIT_SRC", SRC)
    kernels = os.environ["FA_AUDIT_KERNELS"].split(",") if os.environ.get("FA_AUDIT_KERNELS") else KERNELS
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "capi.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-honor-nans", "-S",
               "--cuda-device-only", "-o", out, src] + sys.argv[1:]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
        if os.environ.get("FA_AUDIT_KEEP"):
            open(os.environ["FA_AUDIT_KEEP"], "w").write(text)
    for kernel in kernels:
        audit(text, kernel)
    print("check_owned_regs: ok (no spills, no scratch, no compiler use of accumulator registers)")


def audit(text, KERNEL):
    # the whole function body, up to its end label (a kernel with early exits has several s_endpgm)
    m = re.search(r"^" + KERNEL + r":.*?\n\.Lfunc_end\d+:", text, re.S | re.M)
    if not m:
        sys.exit("check_owned_regs: kernel not found in the assembly")
    problems = []
    in_asm = False
    for line in m.group(0).split("\n"):
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0]
        if not in_asm and (re.search(r"\bv_accvgpr", code) or re.search(r"\ba\[?\d", code)):
            problems.append("compiler-generated accumulator-register access: " + code.strip())
        if re.search(r"\bscratch_(load|store)", code):
            problems.append("scratch access: " + code.strip())
    meta = text[text.find(".amdhsa_kernel " + KERNEL):]
    meta = meta[:meta.find(".end_amdhsa_kernel")]
    for key in (".amdhsa_private_segment_fixed_size",):
        mm = re.search(key + r"\s+(\d+)", meta)
        if mm and int(mm.group(1)) != 0:
            problems.append(f"{key} = {mm.group(1)}")
    lanes = len(re.findall(r"\bv_(?:read|write)lane_b32", m.group(0)))
    if lanes:
        print(f"check_owned_regs: note: {KERNEL} keeps scalar values in vector lanes ({lanes} v_readlane/v_writelane)")
    if problems:
        print("check_owned_regs: FAILED for", KERNEL)
        for p in problems[:20]:
            print("   ", p)
        sys.exit(1)


if __name__ == "__main__":
    main()
