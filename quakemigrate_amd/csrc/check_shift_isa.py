#!/usr/bin/env python3
"""
Build-time check of a register convention the compiler is not told about.

The row-block kernels of qm_shift.hpp (tables of more than 64 rows) keep a 2x2x2 group's 64
accumulators in the generated loop's HARD registers (v80 up, kShiftBlockVgprs) ACROSS separate
inline-asm statements; the asm only lists them as clobbers.  What makes that sound is that the
compiler's own code between two statements (staging, barriers, address arithmetic) never touches
a register from kShiftBlockVgprs up -- obtained with `amdgpu_waves_per_eu(6, 6)` (a budget of 80
registers), which is a heuristic of the current toolchain, not a contract.  This module walks the
ISA the compiler emitted for those kernels and fails if any instruction OUTSIDE the inline-asm
blocks names such a register.  `__graft_entry__.build_engine` runs it on every build of
qm_launch_shift.hip (so a toolchain or flag change cannot silently corrupt sums) and
tests/test_host.py runs it again.

usage: python check_shift_isa.py <qm_launch_shift...gfx950.s> [qm_shift_asm.inc]
"""
import pathlib
import re
import sys

ROW_BLOCK_KERNELS = ("_ZN2qm23stack_shift_rows_kernelILi8EEEvNS_9ShiftArgsE",
                     "_ZN2qm24stack_shift_rows2_kernelILb0ELi8EEEvNS_9ShiftArgsE",
                     "_ZN2qm24stack_shift_rows2_kernelILb1ELi8EEEvNS_9ShiftArgsE",
                     "_ZN2qm24stack_shift_rows4_kernelILb0EEEvNS_9ShiftArgsE",
                     "_ZN2qm24stack_shift_rows4_kernelILb1EEEvNS_9ShiftArgsE")


# round 6: the row-block kernel of the WIDE tiles -- 96 accumulators from v56 up (kShiftWideBlockVgprs,
# amdgpu_waves_per_eu(9, 9))
WIDE_ROW_BLOCK_KERNELS = ("_ZN2qm28stack_shift_wide_rows_kernelENS_9ShiftArgsE",)


def first_hard_register(inc_text, wide=False):
    name = "kShiftWideBlockVgprs" if wide else "kShiftBlockVgprs"
    return int(re.search(name + r" = (\d+);", inc_text).group(1))


def check_all(sasm, inc_text):
    """both families against their own first hard register.  What has to survive between two calls is the
    ACCUMULATORS: from the first hard register up in the 4-sample kernels (everything above is the loop's), the
    96 registers from kShiftWideBlockVgprs in the wide kernel -- its loop names every register up to v255, and the
    compiler may park a value in one of the two or three the eager flavour leaves unnamed (windows, addresses and
    constants are made anew by every call's prologue)."""
    wide_first = first_hard_register(inc_text, wide=True)
    return (check(sasm, first_hard_register(inc_text)) +
            check(sasm, wide_first, WIDE_ROW_BLOCK_KERNELS, last=wide_first + 96))


def check(sasm, first_hard, kernels=ROW_BLOCK_KERNELS, last=1 << 20):
    """Raises AssertionError naming the offending line; returns the number of compiler
    instructions with VGPR operands that were checked."""
    total = 0
    for symbol in kernels:
        assert symbol + ":" in sasm, f"kernel {symbol} not found in the ISA"
        body = sasm[sasm.index(symbol + ":"):]
        body = body[:body.index(".Lfunc_end")]                 # the whole kernel, early exits included
        assert "s_endpgm" in body, symbol
        inside, checked = False, 0
        for line in body.splitlines():
            if ";;#ASMSTART" in line or ";;#ASMEND" in line:
                inside = ";;#ASMSTART" in line
                continue
            if inside:
                continue
            code = line.split(";")[0]
            regs = [int(x) for x in re.findall(r"\bv(\d+)\b", code)]
            regs += [int(b) for _, b in re.findall(r"\bv\[(\d+):(\d+)\]", code)]
            assert all(r < first_hard or r >= last for r in regs), \
                f"{symbol}: compiler code touches a register >= v{first_hard}: {line.strip()}"
            checked += bool(regs)
        assert checked > 100, (symbol, checked)
        total += checked
    return total


if __name__ == "__main__":
    here = pathlib.Path(__file__).resolve().parent
    inc = pathlib.Path(sys.argv[2]) if len(sys.argv) > 2 else here / "qm_shift_asm.inc"
    n = check_all(pathlib.Path(sys.argv[1]).read_text(), inc.read_text())
    print(f"row-block kernels: {n} compiler instructions checked, none touches the asm's registers")
