#!/usr/bin/env python3
"""
Generate qm_ring_asm.inc: the hand-scheduled gfx950 inner loop of the stacking kernel.

One asm statement stacks up to 8 table rows for J samples per lane (64*J samples per
wavefront).  Per row and sample-slot: one ds_read_b64 (the only LDS read form that
runs at the full 256 B/clk/CU on gfx950 besides b128; hipcc otherwise fuses pairs into
ds_read2st64_b64, which runs at half that) and one v_add_f64.  A ring of 8 reads is kept
in flight: steady state is  s_waitcnt lgkmcnt(7) ; v_add_f64 ; ds_read_b64  repeated, so
the LDS queue never drains inside a statement, rows are added in ascending order for every
sample (the order that fixes the float64 rounding, migratelib.c:54-59), and nothing is
in flight across the statement boundary (the compiler never sees a half-landed register).

Usage: python gen_ring_asm.py > qm_ring_asm.inc   (the .inc is committed; build() checks
it is up to date).
"""

import os

IN_FLIGHT = int(os.environ.get("QM_RING_IN_FLIGHT", "8"))     # ds_read_b64 in flight per wave
SETPRIO = int(os.environ.get("QM_RING_SETPRIO", "0"))          # s_setprio level inside the ring
SLOTS = IN_FLIGHT + 1                                          # landing registers (doubles)
IN_FLIGHT32 = int(os.environ.get("QM_RING32_IN_FLIGHT", "8"))  # float32 screening ring
SETPRIO32 = int(os.environ.get("QM_RING32_SETPRIO", "0"))


def variant(J, rows):
    """
    Reads are issued in (row, j) order; read k lands in slot k % SLOTS.  Steady state:
        s_waitcnt lgkmcnt(F-1)      ; read k has landed (DS returns in order)
        ds_read_b64  (read k+F)     ; issued BEFORE the add so VALU contention never delays LDS
        v_add_f64    acc[j] += slot(k)
    Slot (k+F) % (F+1) was last used by read k-1, whose add has already issued.
    """
    kt = 64 * J
    n = rows * J
    F = IN_FLIGHT
    lines = []
    if SETPRIO:
        lines.append(f"s_setprio {SETPRIO}")

    def read(k):
        row, j = divmod(k, J)
        lines.append(f"ds_read_b64 %[t{k % SLOTS}], %[a{row}] offset:{(row * kt + 64 * j) * 8}")

    for k in range(min(F, n)):
        read(k)
    for k in range(n):
        issued_after = min(k + F, n) - 1 - k
        lines.append(f"s_waitcnt lgkmcnt({issued_after})")
        if k + F < n:
            read(k + F)
        lines.append(f"v_add_f64 %[c{k % J}], %[c{k % J}], %[t{k % SLOTS}]")
    if SETPRIO:
        lines.append("s_setprio 0")
    body = "\\n\\t".join(lines)
    outs = [f'[c{j}] "+v"(acc[{j}])' for j in range(J)]
    outs += [f'[t{s}] "=&v"(t[{s}])' for s in range(SLOTS)]
    ins = [f'[a{r}] "v"(addr[{r}])' for r in range(rows)]
    return (f'        asm volatile("{body}"\n'
            f'                     : {", ".join(outs)}\n'
            f'                     : {", ".join(ins)}\n'
            f'                     : "memory");')


def variant32(JP, rows):
    """
    The float32 screening pass (qm_screen.hpp): a lane owns JP pairs of consecutive samples, one
    ds_read_b64 fetches a pair (two floats) and one v_pk_add_f32 adds both.  Row r of a chunk
    starts r * (8 * 128*JP - 8) bytes after the chunk base (two staggered copies of every window,
    see qm_screen.hpp); pair j is 512*j bytes further.  Same ring discipline as above.
    """
    stride = 8 * 128 * JP - 8
    n = rows * JP
    F = IN_FLIGHT32
    SLOTS = F + 1
    lines = []
    if SETPRIO32:
        lines.append(f"s_setprio {SETPRIO32}")

    def read(k):
        row, j = divmod(k, JP)
        lines.append(f"ds_read_b64 %[t{k % SLOTS}], %[a{row}] offset:{row * stride + 512 * j}")

    for k in range(min(F, n)):
        read(k)
    for k in range(n):
        issued_after = min(k + F, n) - 1 - k
        lines.append(f"s_waitcnt lgkmcnt({issued_after})")
        if k + F < n:
            read(k + F)
        lines.append(f"v_pk_add_f32 %[c{k % JP}], %[c{k % JP}], %[t{k % SLOTS}]")
    if SETPRIO32:
        lines.append("s_setprio 0")
    body = "\\n\\t".join(lines)
    outs = [f'[c{j}] "+v"(acc[{j}])' for j in range(JP)]
    outs += [f'[t{s}] "=&v"(t[{s}])' for s in range(SLOTS)]
    ins = [f'[a{r}] "v"(addr[{r}])' for r in range(rows)]
    return (f'        asm volatile("{body}"\n'
            f'                     : {", ".join(outs)}\n'
            f'                     : {", ".join(ins)}\n'
            f'                     : "memory");')


def main32():
    print()
    print("// ---- float32 screening pass: ring32_full<JP> / ring32_tail<JP> (pairs of samples) ----")
    print("typedef float qm_v2f __attribute__((ext_vector_type(2)));")
    print("template <int JP>")
    print("__device__ __forceinline__ void ring32_full(qm_v2f (&acc)[JP], const unsigned (&addr)[8]);")
    print("template <int JP>")
    print("__device__ __forceinline__ void ring32_tail(qm_v2f (&acc)[JP], const unsigned (&addr)[8],"
          " int rows);")
    for JP in (1, 2, 4):
        print()
        print("template <>")
        print(f"__device__ __forceinline__ void ring32_full<{JP}>(qm_v2f (&acc)[{JP}], "
              "const unsigned (&addr)[8]) {")
        print(f"    qm_v2f t[{IN_FLIGHT32 + 1}];")
        print(variant32(JP, 8).replace("        asm", "    asm").replace("                     :", "                 :"))
        print("}")
        print()
        print("template <>")
        print(f"__device__ __forceinline__ void ring32_tail<{JP}>(qm_v2f (&acc)[{JP}], "
              "const unsigned (&addr)[8], int rows) {")
        print(f"    qm_v2f t[{IN_FLIGHT32 + 1}];")
        print("    switch (rows) {")
        for rows in range(7, 0, -1):
            print(f"    case {rows}:")
            print(variant32(JP, rows))
            print("        break;")
        print("    default:")
        print("        break;")
        print("    }")
        print("}")


def main():
    print("// GENERATED by gen_ring_asm.py -- do not edit.  See that file for the schedule.")
    print("// ring_chunk<J>(acc, addr, rows): acc[j] += LDS[addr[r] + (r*64*J + 64*j)*8] for")
    print("// r = 0..rows-1 in ascending r; addr[r] = byte address of the lane's first sample of")
    print("// row r's window MINUS the row's own r*64*J*8 (that part is in the immediates).")
    print("// ring_full<J>: rows = 8 (the hot statement); ring_tail<J>: 1..7 rows.")
    print("template <int J>")
    print("__device__ __forceinline__ void ring_full(double (&acc)[J], const unsigned (&addr)[8]);")
    print("template <int J>")
    print("__device__ __forceinline__ void ring_tail(double (&acc)[J], const unsigned (&addr)[8],"
          " int rows);")
    for J in (1, 2, 4):
        print()
        print("template <>")
        print(f"__device__ __forceinline__ void ring_full<{J}>(double (&acc)[{J}], "
              "const unsigned (&addr)[8]) {")
        print(f"    double t[{SLOTS}];")
        print(variant(J, 8).replace("        asm", "    asm").replace("                     :", "                 :"))
        print("}")
        print()
        print("template <>")
        print(f"__device__ __forceinline__ void ring_tail<{J}>(double (&acc)[{J}], "
              "const unsigned (&addr)[8], int rows) {")
        print(f"    double t[{SLOTS}];")
        print("    switch (rows) {")
        for rows in range(7, 0, -1):
            print(f"    case {rows}:")
            print(variant(J, rows))
            print("        break;")
        print("    default:")
        print("        break;")
        print("    }")
        print("}")


if __name__ == "__main__":
    main()
    main32()
