#!/usr/bin/env python3
"""
Generate qm_shift_asm.inc: the gfx950 inner loop of the shift-reuse stacking kernel.

Idea.  The round-2 kernels fetch every operand of every add from LDS
(8 bytes per add, the binding unit).  Here a lane owns FOUR CONSECUTIVE samples (t = 4*lane + k)
and a wavefront stacks a 2x2x2 GROUP of nodes at a time: for one table row the eight nodes'
delays differ by a few samples, so the operands of all eight nodes are a window of
4 + (largest - smallest delay) consecutive samples per lane.  The window is read ONCE into
registers and node g's four adds take their operands from window registers [idx_g, idx_g + 4)
-- a wave-uniform, data-dependent register index: gfx9 VGPR-index mode (s_set_gpr_idx_on, SRC0
relative).  Per (node, sample) the rows are still added one at a time in ascending order, so
every sum has the reference's bits (migratelib.c:54-59); LDS operand reads drop to about half.

LDS layout of a row window (u = sample index inside the window, staged by the kernel): two
planes of 16-byte slots, plane A slot s = samples (4s, 4s+1), plane B slot s = (4s+2, 4s+3), B
kShiftPlane bytes above A.  Lane l's window quad m (registers 4m .. 4m+3 = samples
e0 + 4l + 4m ..) is then ONE aligned ds_read_b128 per plane at slot e0/4 + l + m: conflict-free,
consecutive registers, half the LDS instructions of 8-byte reads.

Stream format (built once per table, qm_shift.hpp): per (brick, wave) a contiguous run of 32-byte
records: a lead-in record, one record per (group, row) with rows padded to an even count, one
trailing pad.  Record of row r (round 5: every loop reads this "packed" form; QM_SHIFT_PACKED=0
generates the 64-byte form of rounds 3-4 with the eight indices as dwords 0..7, header 8-9, 10-11):
    dwords 0, 1   idx[g] = 2 * (delay_g - e0_r), the register offset of node g's first operand, as
                  byte g of the pair (nodes 0-3 in dword 0, nodes 4-7 in dword 1)
    dwords 2, 3   header of row r + 1 (the lead-in carries row 0's; the last row's is harmless):
                  LDS byte offset of plane A's slot e0/4 of that row; its quad count (2 .. NQMAX)
    dwords 4, 5   row 0 of a group only: flat index of the group's first node, valid-node mask
One s_load_dwordx8 per row, issued one row ahead; the window is read one row ahead into the other
of two register windows (rows unrolled by two).

Round 4 adds two things to the same loop.

TAIL TILES (SPL = 1, 2, 3 samples per lane: time tiles of 64, 128, 192 samples): a scan that is not
a multiple of 256 samples used to compute a whole 256-sample tile for its remainder (401 samples
cost 512, 625 cost 768).  The remainder now runs as ONE tile of 64 * SPL samples with the same
stream and the same register-index scheme: a lane owns SPL consecutive samples, node g's SPL adds
take window registers [idx_g, idx_g + SPL).  Row windows of a tail tile are staged CONTIGUOUSLY
(sample u at byte 8 u of the row's LDS region; the two-plane layout exists to make four samples
per lane conflict-free) -- sample e0 of a row sits at twice the record's plane offset -- and read
with ds_read_b64 (lane stride 8 / 24 bytes) or, for SPL = 2, aligned ds_read_b128 (lane stride 16
bytes): 4 nq - 4 + SPL doubles per row.  A tail tile starts where the full tiles end (nothing is
computed twice); lanes past the scan's end compute on zero-filled / following samples and are
masked where results leave the wavefront (partial sets, per-sample EXEC masks at the volume
stores, zero weights in the marginal sum).

MARGINAL flavour (locate's marginalised map without the 4-D volume, signal/scan.py:720,
io/event.py:421-439): the volume flavour's epilogue with the two stores replaced by
m = sum_k w_k * 2^z_k (w = 1.0 inside the window [m0, m1), 0.0 outside: exact), a wavefront sum
by DPP moves (quad_perm, row_half_mirror, row_mirror, row_bcast 15 / 31: the total lands in lane
63) and one 8-byte store per (node, tile) from that lane.

Round 5: every loop reads 32-byte records (one 64-bit shift per node pair); the row-block flavours ("ONE
group, one block of its rows per call") also STAGE: while a wavefront adds the rows of the block in LDS it
issues the LDS-direct loads that bring its share of the workgroup's NEXT block into the idle half --
stage_step, one row window per pair of rows, scalar bookkeeping only -- and then falls into a plain pair loop.

This file emits the PRODUCT loops only and reads nothing from the environment (round 6):
the constants below are the measured choices.  The timing experiments of rounds 3-5 (loops without their
scalar loads / waits / index switches / window reads, records by pairs, interleaved reads, staging steps
without loads ...: wrong results by construction) lived in this file up to commit e726d9a and are described,
with their numbers, in DESIGN_HISTORY.md and profiles/r0[3-5]_ab_runs.txt; tools/dev/shift_overlay.py patches
the constants of this module for A/B builds of right-result variants (tools/shift_variants.sh).

Usage: python gen_shift_asm.py > qm_shift_asm.inc   (committed; build() checks it is current).
"""

NQMAX = 6                # window = 4 * NQMAX doubles
NQMIN = 4                # quads fetched unconditionally (five or six: C3 +2.2 % / +4.8 %, profiles/r05_ab_runs.txt)
WMAX = 4 * NQMAX
# Record size.  64 bytes: the eight register indices as dwords.  32 bytes ("packed"): as bytes of two
# dwords (nodes 0-3, nodes 4-7) -- half the stream, half the scalar-load bytes, 16 fewer hard SGPRs.
# Round 4 shifted each dword on its own (six more SALU per row): the row-block loops gained 5 % (128
# rows 59.3 -> 56.5 ms), the loops that take all groups of a brick per call lost 1 % (C3 detect 45.8 ->
# 46.3 ms) and kept 64-byte records.  Round 5: the two dwords are ONE 64-bit scalar, shifted once per
# node PAIR (nodes g and g + 4 are added back to back: three s_lshr_b64 per row) -- C3 detect 45.7 vs
# 45.7 ms, the C3 locate volume 5.42 -> 5.26, the C4 slab 176.4 -> 175.0 (profiles/r05_ab_runs.txt):
# every loop now reads 32-byte records, a table's stream is 4 S bytes per node -- the table's own size.
PACKED_GROUPS = True
PACKED_BLOCKS = True
PACKED_SHIFT64 = True    # round 5: see node_adds


def rec_bytes(packed):
    return 32 if packed else 64
PF_AHEAD = 16            # records ahead (0: no prefetch -- C3 +12 %, C4 +17 %; 32 / 64: flat)
NEXT_RUN = True          # round 5: prefetch the head of the wavefront's next run
NEXT_META = False        # ... and the next brick's row-window metadata: flat on the loops that take all groups of a
                         # brick (C3 -0.1 %, C4 -0.2 %: off); the row-block loops always do it (-2.5 %)
PF_EVERY = 2             # 2: one prefetch per PAIR of rows (two 32-byte records: the same 128-byte line either way;
                         # C3 -0.4 %, C4 -0.7 %, locate volume -1.7 %, profiles/r05_ab_runs.txt); 1: per row
STAGING = False          # (set by body(): the pair of rows being emitted carries the staging step)
STAGE_IN_LOOP = True     # row blocks, round 5: see stage_step
PLANE2 = 40896           # plane A -> plane B, bytes (two 4-wave workgroups per CU, 80 KB each): 128 q + 64
                         # keeps the staging stores conflict-free
PLANE3 = 51136           # the same for the 12-wave workgroup (one per CU: 100 KB of windows + 60 KB of
                         # per-wavefront running state)
PLANE8 = 81856           # ... for the 8-wave workgroup that owns a CU's whole LDS (tables of 33-64 rows):
                         # beyond the 16-bit offset of a DS instruction, so plane B gets its own
                         # address register ("far plane")
STATE_CHUNK = 1024       # running state in LDS: 5 chunks of 64 lanes x 16 bytes per wavefront
VB_BLOCK = 80            # first hard VGPR of the row-block flavour
VB_WIDE = 48             # ... of the wide flavours (SPL = 6: 30 VGPRs of running state below it)
VB_WIDE_BLOCK = 56       # ... of the wide row-block flavour (accumulators kept across calls: the compiler's code
                         # between them stays below -- amdgpu_waves_per_eu(9, 9), checked in the ISA at every build)
NQMIN_WIDE = 4           # quads a wide tile's window fetches unconditionally (6: no conditional reads at all)
BUTTERFLY = True         # marginal map: the eight nodes of a whole group summed over the wavefront together
VOLUME_DEGREE = 10       # 2^f of stored values (qm_kernels.hpp: QM_EXP2_DEGREE_VOLUME)
MARGINAL_DEGREE = 8      # 2^f of the marginalised map's terms: the polynomial of the running sums (7.8e-13), every
                         # term positive -- the map inherits at most that (tests: 1e-12)


def configure(lds_state, far=False, lazy=False, block=False, spl=4, contig=False, marginal=False, bmax=False):
    """Register plan.  far: plane B is addressed through a second register (see PLANE8).  lds_state = False: the wavefront's running (max, sum, index) are inline-asm
    operands (20 VGPRs the compiler places below VB).  True: they live in LDS and are read and
    written by the group merge, the per-node temporaries move into window 1 -- 167 VGPRs in all,
    three wavefronts per SIMD."""
    global LDS_STATE, FAR, PLANE, VB, ACC, WIN, VADDR, VADDRB, VNODE, VC, VPF, VZERO, VEND, VMAG
    global F, P, KI, GMAX, GIDX, TT, GSUM, MAXR, SUMR, IDXR, LAZY, BLOCK, SPL, CONTIG, MARGINAL, VDUMMY, BMAX
    BLOCK = block            # one group per call, accumulators kept between calls (row blocks)
    BMAX = bmax              # groups whose largest z comes within the tie slack of the wavefront's running maximum
                             # also raise the BRICK's row of maxima in global memory (tie_rule = 1, see brick_max)
    assert not bmax or (spl == 6 and not block and not lds_state)
    configure_scalars(PACKED_BLOCKS if block else PACKED_GROUPS)
    global AS
    SPL = spl                # samples per lane (4: full tiles; 1..3: tail tiles; 6: WIDE tiles of 384 samples)
    CONTIG = contig          # row windows staged contiguously (tail tiles, wide tiles)
    MARGINAL = marginal      # the marginalised map instead of the volume
    wide = spl == 6
    assert (1 <= spl <= 4 or wide) and (contig or spl == 4)
    assert not (contig and (far or lds_state)) and not (contig and (lazy or block) and not wide)
    assert not (wide and marginal)
    AS = 2 * spl if wide else 8      # VGPRs between the accumulators of consecutive nodes
    LDS_STATE = lds_state
    LAZY = lazy
    FAR = far
    PLANE = PLANE3 if lds_state else PLANE8 if far else PLANE2
    # first hard VGPR (row-block flavour: the accumulators must survive the compiler's code between
    # two calls, which therefore has to stay below VB -- tests/test_host.py checks the ISA)
    VB = 4 if lds_state else VB_WIDE_BLOCK if block and wide else VB_BLOCK if block else VB_WIDE if wide else 32
    ACC = VB                 # acc[g][k] = v[ACC + AS g + 2 k : +1]
    WIN = [ACC + 8 * AS, ACC + 8 * AS + 2 * WMAX]
    VADDR = WIN[1] + 2 * WMAX
    VADDRB = VADDR + 1 if far else VADDR     # plane B's address register (far plane only)
    # epilogue temporaries live in window 1 (free between a group's last row and the next group's
    # row 1)
    TS = 2 * spl if wide else 8      # (wide: four sets of six pairs fill window 1, the group's indices follow VMAG)
    F = WIN[1]
    P = WIN[1] + TS
    TT = WIN[1] + 2 * TS     # z + magic (SPL pairs)
    GMAX = WIN[1] + 3 * TS
    GIDX = WIN[1] + 32
    GSUM = WIN[1] + 36       # (LDS state only) the group's sum of 2^z
    KI = WIN[1] + 44         # "no index"
    if lds_state:
        VNODE = WIN[1] + 45
        VC = WIN[1] + 46     # leading polynomial coefficient (pair), re-made per epilogue
        VPF = VADDRB + 1     # L2 prefetch of the stream: dummy destination, zero offset
        assert VC + 2 <= WIN[1] + 2 * WMAX
        # at the merge F / P / TT are dead: the state read from LDS lands there
        MAXR = [v2(F + 2 * k) for k in range(4)]
        SUMR = [v2(GSUM + 2 * k) for k in range(4)]
        IDXR = [f"v{TT + k}" for k in range(4)]
    else:
        VNODE = VADDRB + 1
        VC = (VNODE + 2) & ~1    # (64-bit register tuples are even-aligned on gfx90a+)
        VPF = VC + 2
        assert KI + 1 <= WIN[1] + 2 * WMAX
        MAXR = [f"%[max{k}]" for k in range(SPL)]
        SUMR = [f"%[sum{k}]" for k in range(SPL)]
        IDXR = [f"%[idx{k}]" for k in range(SPL)]
    VZERO = VPF + 1
    VMAG = (VZERO + 2) & ~1      # 1.5 * 2^52 as a VGPR pair (lazy flavour: z is folded into FMAs)
    VEND = VMAG + 2 if lazy else VZERO + 1
    if wide and block:
        # (one group per call: no next group's window is on its way into window 0 while the epilogue runs)
        GIDX = WIN[0]
        KI = WIN[0] + SPL
        assert VEND <= 256
    elif wide:
        GIDX = VEND
        KI = VEND + SPL
        VEND = KI + 1
        assert GMAX + 2 * SPL <= WIN[1] + 2 * WMAX and VEND <= 256


SB = 48                  # first hard SGPR (s_load_dwordx16 / x8 destinations)


def configure_scalars(packed):
    """hard SGPR plan; the record buffers shrink with packed records"""
    global PACKED, REC, NBUF, BUF, R_HDR, R_BASE, ST, SBASE, SMASK, SPAIRS, SNODE, STAB, SOFF, SPF, SVA
    global SNEGINF, SMAGIC, SEND, SG_META, SG_ROW, SG_AFTER, SG_SRC, SG_T, SG_LDS
    PACKED = packed
    REC = rec_bytes(packed)  # bytes per stream record
    NBUF = REC // 4          # dwords of a record
    BUF = [SB, SB + NBUF]
    R_HDR = 2 if packed else 8     # dwords of the next row's header inside a record (LDS offset, quad count)
    R_BASE = 4 if packed else 10   # ... of the group's first node and valid-node mask (row 0 of a group)
    ST = SB + 2 * NBUF       # [ST:ST+1], [ST+2:ST+3] compare masks
    SBASE = ST + 4
    SMASK = ST + 5
    SPAIRS = ST + 6
    SNODE = ST + 7
    STAB = ST + 8            # stream base (pair, even)
    SOFF = ST + 10           # byte offset of the record loaded last
    SPF = ST + 12            # prefetch address (pair)
    SVA = ST + 14            # volume row address of the node in the epilogue (pair)
    SNEGINF = ST + 16        # -inf (pair)
    SMAGIC = ST + 18         # 1.5 * 2^52 (pair): z + magic has rint(z) in its low dword
    SEND = ST + 20
    if BLOCK and STAGE_IN_LOOP:
        # row blocks, round 5: the NEXT block's staging is issued from inside the row loop (stage_step)
        SG_META = (SEND + 3) & ~3     # the staging row's window record (min delay, span, first slot, slots)
        SG_ROW = SG_META + 4          # the row this wavefront stages next
        SG_AFTER = SG_META + 5        # vector loads of its own the call has issued since its last staging load
        SG_SRC = SG_META + 6          # (pair) the chunk's global address
        SG_T = SG_META + 8            # (pair) temporaries
        SG_LDS = SG_META + 10         # LDS address of the row's first slot, plane A
        SEND = SG_META + 12
    assert STAB % 2 == 0 and SB % 4 == 0 and SVA % 2 == 0 and SPF % 2 == 0 and SNEGINF % 2 == 0 and SEND <= 102


def v2(r):
    return f"v[{r}:{r + 1}]"


def s2(r):
    return f"s[{r}:{r + 1}]"


class Emitter:
    def __init__(self):
        self.lines = []
        self.nlabel = 0

    def __call__(self, text):
        self.lines.append(text)

    def label(self, stem):
        self.nlabel += 1
        return f"L{stem}{self.nlabel}_%="


def quad_reads(win, m):
    plane_b = f"v{VADDRB} offset:{16 * m}" if FAR else f"v{VADDR} offset:{PLANE + 16 * m}"
    return [f"ds_read_b128 v[{win + 8 * m}:{win + 8 * m + 3}], v{VADDR} offset:{16 * m}",
            f"ds_read_b128 v[{win + 8 * m + 4}:{win + 8 * m + 7}], {plane_b}"]


def contig_reads(win, m):
    """tail tiles: the reads quad count m + 1 adds to quad count m (m < NQMIN: nothing, the first
    NQMIN quads' worth -- 4 NQMIN - 4 + SPL doubles -- is fetched by contig_base_reads).  Wide tiles: a quad
    count of nq stands for a window of 4 nq doubles = 2 nq aligned pairs (the window starts at an EVEN sample:
    lane stride 48 bytes, 16-byte reads, conflict-free: 3 l mod 16 is a permutation of the 16 lanes of every
    ds_read_b128 lane group)"""
    if SPL == 6:
        return [f"ds_read_b128 v[{win + 4 * p}:{win + 4 * p + 3}], v{VADDR} offset:{16 * p}"
                for p in (2 * m, 2 * m + 1)]
    if SPL == 2:             # pairs of doubles, 16-byte aligned: 2 nq - 1 of them
        return [f"ds_read_b128 v[{win + 4 * p}:{win + 4 * p + 3}], v{VADDR} offset:{16 * p}"
                for p in range(2 * m - 1, 2 * m + 1)]
    return [f"ds_read_b64 {v2(win + 2 * i)}, v{VADDR} offset:{8 * i}"
            for i in range(4 * m - 4 + SPL, 4 * m + SPL)]


def contig_base_reads(win):
    if SPL == 6:
        return [f"ds_read_b128 v[{win + 4 * p}:{win + 4 * p + 3}], v{VADDR} offset:{16 * p}"
                for p in range(2 * NQMIN_WIDE)]
    if SPL == 2:
        return [f"ds_read_b128 v[{win + 4 * p}:{win + 4 * p + 3}], v{VADDR} offset:{16 * p}"
                for p in range(2 * NQMIN - 1)]
    return [f"ds_read_b64 {v2(win + 2 * i)}, v{VADDR} offset:{8 * i}"
            for i in range(4 * NQMIN - 4 + SPL)]


def window_address(e, hdr):
    if CONTIG:
        # sample e0 of the row: byte 2 * (the record's plane offset) of the contiguous layout
        e(f"v_lshl_add_u32 v{VADDR}, s{hdr}, 1, %[lane]")
        return
    e(f"v_add_u32 v{VADDR}, s{hdr}, %[lane]")          # src0 scalar: untouched by SRC0-relative mode
    if FAR:
        e(f"v_add_u32 v{VADDRB}, s{hdr}, %[laneb]")


def issue_window(e, q, hdr):
    """block form (prologue only): reads of the row whose header is s[hdr], s[hdr+1] into WIN[q]"""
    window_address(e, hdr)
    if CONTIG:
        for line in contig_base_reads(WIN[q]):
            e(line)
    else:
        for m in range(NQMIN):
            for line in quad_reads(WIN[q], m):
                e(line)
    done = e.label("rd")
    for m in range(NQMIN_WIDE if SPL == 6 else NQMIN, NQMAX):
        e(f"s_cmp_le_u32 s{hdr + 1}, {m}")
        e(f"s_cbranch_scc1 {done}")
        for line in (contig_reads(WIN[q], m) if CONTIG else quad_reads(WIN[q], m)):
            e(line)
    e(f"{done}:")


def node_order():
    """order in which a row's eight nodes are added (each has its own accumulators: any order gives the
    same sums).  Packed records: nodes g and g + 4 sit in the same byte of the record's two index
    dwords, so they are taken together and ONE 64-bit shift moves both dwords to the next pair."""
    return (0, 4, 1, 5, 2, 6, 3, 7) if PACKED and PACKED_SHIFT64 else tuple(range(8))


def node_adds(e, p, g, first, rec=None):
    first = first and not BLOCK          # (row blocks: the accumulators are zeroed, or carry on)
    rec = BUF[p] if rec is None else rec
    if PACKED:
        # the record packs the eight indices as bytes of two dwords (nodes 0-3, nodes 4-7); the
        # instruction takes bits [7:0] of its operand, so nodes 0 and 4 use the dwords as they
        # are; round 5: the pair of dwords is shifted as ONE 64-bit scalar before nodes (1, 5),
        # (2, 6), (3, 7) -- three SALU instructions per row instead of six
        reg = rec + g // 4
        if PACKED_SHIFT64:
            if g in (1, 2, 3):
                e(f"s_lshr_b64 {s2(rec)}, {s2(rec)}, 8")
        elif g % 4:
            e(f"s_lshr_b32 s{reg}, s{reg}, 8")
        e(f"s_set_gpr_idx_on s{reg}, 1")
    else:
        e(f"s_set_gpr_idx_on s{BUF[p] + g}, 1")           # SRC0 relative, index = idx[g]
    for k in range(SPL):
        a = v2(ACC + AS * g + 2 * k)
        w = v2(WIN[p] + 2 * k)
        e(f"v_add_f64 {a}, {w}, {'0' if first else a}")


def stage_step(e):
    """Row blocks: stage ONE row window of the workgroup's next block (LDS-direct loads: global memory -> LDS
    at M0 + 16 lane, no registers) from inside the row loop, one row per pair of rows added.  Round 5: issued as
    a burst in front of the loop -- 20 loads per wavefront, all eight wavefronts at once -- the loads queued up
    in the CU's one vector-memory path and every wavefront stood still until its own were accepted: 15 % of a
    128-row step (the loop without any staging: 55.1 -> 44.9 ms, profiles/r05_ab_runs.txt).  Spread over the
    block they cost their issue slots.  Everything is wave-uniform: the row's record (s_load_dwordx4 one step
    ahead) gives the window's first sample, first slot and slot count; slots 0-63 and 64-(slots - 1) of both
    planes are four loads with the row pointer as their scalar base and lane * 32 as offset (plane B: the
    instruction offset 16 moves the global AND the LDS address, so its M0 is 16 short); EXEC masks the second
    chunk.  M0 is shared with the register-index mode: the step sits between two nodes, the next one's
    s_set_gpr_idx_on rewrites the bits the mode reads.  The kernel hands over rows only when every window of
    the table lies inside the onsets' rows for this tile (no partial slots) and holds at most 127 slots."""
    skip = e.label("sg")
    m, t = SG_META, SG_T
    e(f"s_cmp_ge_u32 s{SG_ROW}, %[stgn]")
    e(f"s_cbranch_scc1 {skip}")
    if CONTIG:
        # wide tiles: ONE contiguous window per row, slots of 32 bytes: pairs 0-191 (384 samples: every window
        # holds them) are three full loads, the rest (up to 62 pairs: the kernel hands rows over only while no
        # window holds more than 127 slots) a fourth under EXEC; the instruction offset moves the global and the
        # LDS address together, so M0 is set once
        e(f"s_mul_i32 s{SG_SRC}, s{SG_ROW}, %[stgt8]")
        e(f"s_mul_hi_u32 s{SG_SRC + 1}, s{SG_ROW}, %[stgt8]")
        e(f"s_ashr_i32 s{t + 1}, s{m}, 31")
        e(f"s_mov_b32 s{t}, s{m}")
        e(f"s_lshl_b64 {s2(t)}, {s2(t)}, 3")
        e(f"s_add_u32 s{SG_SRC}, s{SG_SRC}, s{t}")
        e(f"s_addc_u32 s{SG_SRC + 1}, s{SG_SRC + 1}, s{t + 1}")
        e(f"s_add_u32 s{SG_SRC}, s{SG_SRC}, %[stglo]")
        e(f"s_addc_u32 s{SG_SRC + 1}, s{SG_SRC + 1}, %[stghi]")
        e(f"s_lshl_b32 s{SG_LDS}, s{m + 2}, 5")
        e(f"s_add_u32 m0, s{SG_LDS}, %[stglds]")
        e("s_nop 0")
        for c in range(3):
            e(f"global_load_lds_dwordx4 %[lane32], {s2(SG_SRC)}" + (f" offset:{1024 * c}" if c else ""))
        e(f"s_lshl_b32 s{t}, s{m + 3}, 1")                          # pairs of the window
        e(f"s_sub_u32 s{t}, s{t}, 192")
        e(f"s_bfm_b64 exec, s{t}, 0")
        e(f"global_load_lds_dwordx4 %[lane32], {s2(SG_SRC)} offset:3072")
        e("s_mov_b64 exec, -1")
        e(f"s_mov_b32 s{SG_AFTER}, 0")
        e(f"s_add_u32 s{SG_ROW}, s{SG_ROW}, %[stgstride]")
        e(f"{skip}:")
        return
    e(f"s_mul_i32 s{SG_SRC}, s{SG_ROW}, %[stgt8]")              # row * bytes per onset row (64 bits)
    e(f"s_mul_hi_u32 s{SG_SRC + 1}, s{SG_ROW}, %[stgt8]")
    e(f"s_ashr_i32 s{t + 1}, s{m}, 31")                         # + 8 * (the window's first sample)
    e(f"s_mov_b32 s{t}, s{m}")
    e(f"s_lshl_b64 {s2(t)}, {s2(t)}, 3")
    e(f"s_add_u32 s{SG_SRC}, s{SG_SRC}, s{t}")
    e(f"s_addc_u32 s{SG_SRC + 1}, s{SG_SRC + 1}, s{t + 1}")
    e(f"s_add_u32 s{SG_SRC}, s{SG_SRC}, %[stglo]")
    e(f"s_addc_u32 s{SG_SRC + 1}, s{SG_SRC + 1}, %[stghi]")
    e(f"s_lshl_b32 s{SG_LDS}, s{m + 2}, 4")
    e(f"s_add_u32 s{SG_LDS}, s{SG_LDS}, %[stglds]")
    e(f"s_mov_b32 m0, s{SG_LDS}")
    e("s_nop 0")
    e(f"global_load_lds_dwordx4 %[lane32], {s2(SG_SRC)}")
    e(f"s_add_u32 m0, s{SG_LDS}, {PLANE - 16}")
    e("s_nop 0")
    e(f"global_load_lds_dwordx4 %[lane32], {s2(SG_SRC)} offset:16")
    e(f"s_sub_u32 s{t}, s{m + 3}, 64")                          # slots 64 ..: 0-63 of them
    e(f"s_bfm_b64 exec, s{t}, 0")
    e(f"s_add_u32 s{SG_SRC}, s{SG_SRC}, 2048")
    e(f"s_addc_u32 s{SG_SRC + 1}, s{SG_SRC + 1}, 0")
    e(f"s_add_u32 m0, s{SG_LDS}, 1024")
    e("s_nop 0")
    e(f"global_load_lds_dwordx4 %[lane32], {s2(SG_SRC)}")
    e(f"s_add_u32 m0, s{SG_LDS}, {1024 + PLANE - 16}")
    e("s_nop 0")
    e(f"global_load_lds_dwordx4 %[lane32], {s2(SG_SRC)} offset:16")
    e("s_mov_b64 exec, -1")
    e(f"s_mov_b32 s{SG_AFTER}, 0")
    e(f"s_add_u32 s{SG_ROW}, s{SG_ROW}, %[stgstride]")
    e(f"{skip}:")


def stage_record(e):
    """the record of the row the wavefront stages next (the buffer has slack past its end).  Issued at a row's
    top beside the stream's: in the step itself, two thirds into a row, it was still on its way to L2 and
    back at the next row's top (every wait there is a wait for everything) -- 9 % of a 128-row step."""
    e(f"s_lshl_b32 s{SG_T}, s{SG_ROW}, 4")
    e(f"s_load_dwordx4 s[{SG_META}:{SG_META + 3}], %[stgmeta], s{SG_T}")


def row_iter(e, p, first):
    """row of parity p: its record is in BUF[p], its window in WIN[p].  The next row's window is
    requested before this row's adds (a block of reads at the row's start: spread between the adds,
    the late ones land after the next row's wait -- measured slower, tools/micro results r03a; round 6, wide
    flavours, ONE read behind each of the row's first ten adds, the last of them thirty adds before the row's
    end: 42.54 against 42.49 ms at C3, profiles/r06_ab_runs.txt -- a read costs the SIMD its four cycles
    wherever it is issued)."""
    q = 1 - p
    hdr = BUF[p] + R_HDR
    e("s_waitcnt lgkmcnt(0)")                                  # both have landed
    if first:
        e(f"s_mov_b32 s{SBASE}, s{BUF[p] + R_BASE}")
        e(f"s_mov_b32 s{SMASK}, s{BUF[p] + R_BASE + 1}")
    e(f"s_add_u32 s{SOFF}, s{SOFF}, {REC}")
    e(f"s_load_dwordx{NBUF} s[{BUF[q]}:{BUF[q] + NBUF - 1}], {s2(STAB)}, s{SOFF}")
    if p == 0 and STAGING:
        stage_record(e)
    issue_window(e, q, hdr)
    for pos, g in enumerate(node_order()):
        node_adds(e, p, g, first)
        if pos == 3 and PF_AHEAD and (PF_EVERY == 1 or p == 0):
            # pull the record PF_AHEAD rows ahead into L2 with a vector load nobody waits for (its
            # own counter, vmcnt): the scalar load then finds it there instead of in HBM
            e(f"global_load_dword v{VPF}, v{VZERO}, {s2(SPF)}")
            e(f"s_add_u32 s{SPF}, s{SPF}, {REC * (2 if PF_EVERY == 2 else 1)}")
            e(f"s_addc_u32 s{SPF + 1}, s{SPF + 1}, 0")
            if BLOCK and STAGE_IN_LOOP and (STAGING or FAR):
                e(f"s_add_u32 s{SG_AFTER}, s{SG_AFTER}, 1")
        if pos == 5 and p == 1 and STAGING:
            stage_step(e)


def node_index(e, g, to_vgpr=True):
    # flat index of node g: base + dx*ny*nz + dy*nz + dz (g = 4 dx + 2 dy + dz)
    e(f"s_mov_b32 s{SNODE}, s{SBASE}")
    if g & 4:
        e(f"s_add_u32 s{SNODE}, s{SNODE}, %[nynz]")
    if g & 2:
        e(f"s_add_u32 s{SNODE}, s{SNODE}, %[nz]")
    if g & 1:
        e(f"s_add_u32 s{SNODE}, s{SNODE}, 1")
    if to_vgpr:
        e(f"v_mov_b32 v{VNODE}, s{SNODE}")


def node_tail(e, g, A, opens_group):
    """the node's terms into the running sums and its arg-max (independent of what precedes)"""
    if not (LDS_STATE and opens_group):
        for k in range(SPL):
            e(f"v_add_f64 {SUMR[k]}, {SUMR[k]}, {v2(P + 2 * k)}")
    if LAZY:
        if opens_group:
            return                                             # node 1 takes max(z0, z1)
        for k in range(SPL):
            prev = v2(ACC + 2 * k) if opens_group is None else v2(GMAX + 2 * k)
            e(f"v_max_f64 {v2(GMAX + 2 * k)}, {prev}, {v2(A[k])}")       # a NaN never wins
        return
    for k in range(SPL):
        if opens_group:
            # the group's first node against the (-inf, none) start, without materialising it
            # (a NaN or -inf z leaves (-inf, none), exactly as the strict '>' below would)
            e(f"v_cmp_gt_f64 vcc, {v2(A[k])}, {s2(SNEGINF)}")
            e(f"v_cndmask_b32 v{GIDX + k}, v{KI}, v{VNODE}, vcc")   # v[KI] = "none"
            e(f"v_max_f64 {v2(GMAX + 2 * k)}, {v2(A[k])}, {s2(SNEGINF)}")
        else:
            e(f"v_cmp_gt_f64 vcc, {v2(A[k])}, {v2(GMAX + 2 * k)}")
            e(f"v_cndmask_b32 v{GIDX + k}, v{GIDX + k}, v{VNODE}, vcc")
            e(f"v_max_f64 {v2(GMAX + 2 * k)}, {v2(GMAX + 2 * k)}, {v2(A[k])}")


def epilogue_node(e, degree, volume, g, opens_group, defer=False):
    """LAZY flavour: only the group's maximum is kept per node (v_max_f64); WHICH node holds it is
    recovered after the eight nodes, per sample slot, and only where the group's maximum reaches
    the wavefront's running one (epilogue) -- one instead of three instructions per node-sample
    for the arg-max where that is rare: a lane's running maximum changes about ln(n) times in n
    groups, so the flavour pays for wavefronts that see hundreds of groups (C3: 549, -2 %) and
    costs where they see a few dozen (C1: 48, +1 %); the kernel takes it from kShiftLazyGroups
    on."""
    if volume or not LAZY:
        node_index(e, g, not LAZY)
    A = [ACC + AS * g + 2 * k for k in range(SPL)]
    if LAZY:
        # z = stack * scale is never formed per node-sample: t = fma(stack, scale, 1.5*2^52),
        # k = t - 1.5*2^52, f = fma(stack, scale, -k) (three instructions instead of four; the sum's
        # terms differ from the other kernels' in the last bits, the maximum is taken over the raw
        # stacks -- monotone in z -- and brought to z where a group is examined, epilogue)
        for k in range(SPL):
            e(f"v_fma_f64 {v2(TT + 2 * k)}, {v2(A[k])}, %[scale], {v2(VMAG)}")
        for k in range(SPL):
            e(f"v_add_f64 {v2(F + 2 * k)}, {v2(TT + 2 * k)}, -{s2(SMAGIC)}")
        for k in range(SPL):
            e(f"v_fma_f64 {v2(F + 2 * k)}, {v2(A[k])}, %[scale], -{v2(F + 2 * k)}")
    else:
        for k in range(SPL):                                      # z = stack * log2(e)/available
            e(f"v_mul_f64 {v2(A[k])}, {v2(A[k])}, %[scale]")
        # k = rint(z), f = z - k:  t = z + 1.5*2^52 holds k in its low dword (round half to even, as
        # v_rndne_f64), t - 1.5*2^52 is k as a double -- two adds instead of rndne + the slower cvt
        for k in range(SPL):
            e(f"v_add_f64 {v2(TT + 2 * k)}, {v2(A[k])}, {s2(SMAGIC)}")
        for k in range(SPL):
            e(f"v_add_f64 {v2(F + 2 * k)}, {v2(TT + 2 * k)}, -{s2(SMAGIC)}")
        for k in range(SPL):                                      # f = z - k
            e(f"v_add_f64 {v2(F + 2 * k)}, {v2(A[k])}, -{v2(F + 2 * k)}")
    for k in range(SPL):
        e(f"v_fma_f64 {v2(P + 2 * k)}, {v2(VC)}, {v2(F + 2 * k)}, %[c{degree - 1}]")
    for i in range(degree - 2, -1, -1):
        for k in range(SPL):
            e(f"v_fma_f64 {v2(P + 2 * k)}, {v2(P + 2 * k)}, {v2(F + 2 * k)}, %[c{i}]")
    for k in range(SPL):
        dst = SUMR[k] if (LDS_STATE and opens_group) else v2(P + 2 * k)   # the group's sum starts here
        e(f"v_ldexp_f64 {dst}, {v2(P + 2 * k)}, v{TT + 2 * k}")
    if volume and MARGINAL and defer:
        # whole groups: the lane's share m_g = sum_k w_k * 2^z_k of node g goes into the node's first
        # accumulator pair (dead once its z has been compared); the eight are summed over the
        # wavefront together after the last node (marginal_butterfly)
        node_tail(e, g, A, opens_group)
        dst = v2(ACC + 8 * g)
        e(f"v_mul_f64 {dst if SPL == 1 else v2(F)}, {v2(P)}, %[w0]")
        for k in range(1, SPL):
            e(f"v_fma_f64 {dst if k == SPL - 1 else v2(F)}, {v2(P + 2 * k)}, %[w{k}], {v2(F)}")
        return
    if volume and MARGINAL:
        # (groups cut by the grid's edge: node by node)
        # the marginalised map: m = sum_k w_k * 2^z_k over the lane's samples (w = 1.0 inside the
        # window and the tile's own samples, 0.0 elsewhere: products and the first sum are exact),
        # summed over the wavefront with DPP moves in the order of an xor butterfly (as
        # wave_sum_to_last_lane, qm_kernels.hpp: the total lands in lane 63), one 8-byte store per
        # (node, tile).  A DPP read needs 2 wait states after the VALU write of its source: the
        # node's remaining, independent instructions (its terms into the running sums, its
        # arg-max) are dealt into those gaps, s_nop where they run out.
        # F is dead after the Horner steps: m in F[0:1], the moved copy in F[2:3].
        assert not LAZY and not LDS_STATE
        pool = Emitter()
        node_tail(pool, g, A, opens_group)
        pool = pool.lines

        def gap():
            take, pool[:] = pool[:2], pool[2:]
            # (a v_cmp and the v_cndmask that reads its vcc stay in order: nothing here writes vcc)
            for line in take:
                e(line)
            if len(take) < 2:
                e(f"s_nop {1 - len(take)}")
        m, t = F, F + 2
        e(f"v_mul_f64 {v2(m)}, {v2(P)}, %[w0]")
        for k in range(1, SPL):
            e(f"v_fma_f64 {v2(m)}, {v2(P + 2 * k)}, %[w{k}], {v2(m)}")
        for ctrl in ("quad_perm:[1,0,3,2] row_mask:0xf", "quad_perm:[2,3,0,1] row_mask:0xf",
                     "row_half_mirror row_mask:0xf", "row_mirror row_mask:0xf",
                     "row_bcast:15 row_mask:0xa", "row_bcast:31 row_mask:0xc"):
            gap()
            e(f"v_mov_b32_dpp v{t}, v{m} {ctrl} bank_mask:0xf")
            e(f"v_mov_b32_dpp v{t + 1}, v{m + 1} {ctrl} bank_mask:0xf")
            e(f"v_add_f64 {v2(m)}, {v2(m)}, {v2(t)}")
        e(f"s_lshl_b32 s{SVA}, s{SNODE}, 3")                  # byte offset of the node's element
        e(f"s_lshr_b32 s{SVA + 1}, s{SNODE}, 29")
        e(f"s_add_u32 s{SVA}, s{SVA}, %[vlo]")
        e(f"s_addc_u32 s{SVA + 1}, s{SVA + 1}, %[vhi]")
        e("s_mov_b32 exec_lo, 0")
        e("s_mov_b32 exec_hi, 0x80000000")                     # lane 63
        e(f"global_store_dwordx2 v{VZERO}, {v2(m)}, {s2(SVA)}")
        e("s_mov_b64 exec, -1")
        for line in pool:
            e(line)
        return
    elif volume and CONTIG:
        # tail tile: the lane's SPL values are 8 SPL contiguous bytes of the node's volume row;
        # one 8-byte store per sample slot, each under the mask of the lanes whose sample k lies
        # inside the scan (the tile may reach past its end)
        e(f"s_mul_i32 s{SVA}, s{SNODE}, %[vstride]")
        e(f"s_mul_hi_u32 s{SVA + 1}, s{SNODE}, %[vstride]")
        e(f"s_add_u32 s{SVA}, s{SVA}, %[vlo]")
        e(f"s_addc_u32 s{SVA + 1}, s{SVA + 1}, %[vhi]")
        for k in range(SPL):
            e(f"s_mov_b64 exec, %[m{k}]")
            e(f"global_store_dwordx2 %[voff], {v2(P + 2 * k)}, {s2(SVA)} offset:{8 * k} nt")
        e("s_mov_b64 exec, -1")
    elif volume:
        # the node's four values per lane are 32 contiguous bytes of its volume row (the tile's
        # first sample is in the base): two 16-byte stores.  P is not written again before the
        # next node's first Horner step, a dozen instructions away (gfx940+: 2 wait states
        # between a store of more than 8 bytes and a VALU write to its data registers)
        e(f"s_mul_i32 s{SVA}, s{SNODE}, %[vstride]")
        e(f"s_mul_hi_u32 s{SVA + 1}, s{SNODE}, %[vstride]")
        e(f"s_add_u32 s{SVA}, s{SVA}, %[vlo]")
        e(f"s_addc_u32 s{SVA + 1}, s{SVA + 1}, %[vhi]")
        # (re-dealing the dwords inside each quad of lanes with DPP moves so that one store
        # writes whole 64-byte lines was measured too: same 5.95 ms -- the cost of the stores is
        # their issue inside the CU, profiles/r03_ab_runs.txt)
        # lanes whose four samples the previous tile has already stored (a last tile pulled
        # back over its predecessor) are masked off: no HBM byte is written twice
        e("s_mov_b32 exec_lo, %[mlo]")
        e("s_mov_b32 exec_hi, %[mhi]")
        e(f"global_store_dwordx4 %[voff], v[{P}:{P + 3}], {s2(SVA)} nt")
        e(f"global_store_dwordx4 %[voff], v[{P + 4}:{P + 7}], {s2(SVA)} offset:16 nt")
        e("s_mov_b64 exec, -1")
    node_tail(e, g, A, opens_group)


def marginal_butterfly(e):
    """Wavefront sums of the eight nodes' shares M_g = v[ACC + 8 g : +1] at once: three exchange levels
    in which a lane keeps half of its values and hands the other half to its partner (row_half_mirror,
    then the two quad permutations; the partner keeps the complementary half), so that 8 -> 4 -> 2 -> 1
    values per lane remain, each the sum over 2 -> 4 -> 8 lanes; then the eight-lane sums are added
    across the wavefront (row_ror:8, two ds_bpermute for the rows 16 and 32 lanes away).  Lane l ends
    with the total of node 4 b2 + 2 b0 + b1 of its lane index; lanes 0..7 store.  49 + 5 VALU
    instructions per group instead of 8 x 18 for eight separate sums.  Temporaries: F, P, TT (dead
    after the last node)."""
    def pair_sel(dst, mask, a_if0, a_if1):              # dst = mask ? a_if1 : a_if0 (64-bit)
        for h in range(2):
            e(f"v_cndmask_b32 v{dst + h}, v{a_if0 + h}, v{a_if1 + h}, {mask}")

    def pair_dpp(dst, src, ctrl):
        for h in range(2):
            e(f"v_mov_b32_dpp v{dst + h}, v{src + h} {ctrl} row_mask:0xf bank_mask:0xf")

    Mg = [ACC + 8 * g for g in range(8)]
    # level 1: partner 7 - i within 8 lanes; lanes with bit 2 keep nodes 4..7
    for j in range(4):
        pair_sel(P + 2 * j, "%[mb2]", Mg[4 + j], Mg[j])            # send: b2 ? M_j : M_{4+j}
    for j in range(4):
        pair_sel(F + 2 * j, "%[mb2]", Mg[j], Mg[4 + j])            # keep: b2 ? M_{4+j} : M_j
    for j in range(4):
        pair_dpp(TT + 2 * j, P + 2 * j, "row_half_mirror")
    for j in range(4):
        e(f"v_add_f64 {v2(F + 2 * j)}, {v2(F + 2 * j)}, {v2(TT + 2 * j)}")     # X_j
    # level 2: partner i ^ 1; lanes with bit 0 keep X_2, X_3
    for j in range(2):
        pair_sel(P + 2 * j, "%[mb0]", F + 4 + 2 * j, F + 2 * j)    # send: b0 ? X_j : X_{2+j}
    for j in range(2):
        pair_sel(P + 4 + 2 * j, "%[mb0]", F + 2 * j, F + 4 + 2 * j)   # keep: b0 ? X_{2+j} : X_j
    for j in range(2):
        pair_dpp(TT + 2 * j, P + 2 * j, "quad_perm:[1,0,3,2]")
    for j in range(2):
        e(f"v_add_f64 {v2(P + 2 * j)}, {v2(P + 4 + 2 * j)}, {v2(TT + 2 * j)}")  # Y_j
    # level 3: partner i ^ 2; lanes with bit 1 keep Y_1
    pair_sel(F + 2, "%[mb1]", P + 2, P)                             # send: b1 ? Y_0 : Y_1
    pair_sel(F, "%[mb1]", P, P + 2)                                 # keep: b1 ? Y_1 : Y_0
    pair_dpp(TT, F + 2, "quad_perm:[2,3,0,1]")
    e(f"v_add_f64 {v2(F)}, {v2(F)}, {v2(TT)}")                       # the node's sum over 8 lanes
    # across the wavefront: the other 8 lanes of the row, then the rows 16 and 32 lanes away
    e("s_nop 1")
    pair_dpp(TT, F, "row_ror:8")
    e(f"v_add_f64 {v2(F)}, {v2(F)}, {v2(TT)}")
    for addr in ("%[x16]", "%[x32]"):
        e(f"ds_bpermute_b32 v{TT}, {addr}, v{F}")
        e(f"ds_bpermute_b32 v{TT + 1}, {addr}, v{F + 1}")
        e("s_waitcnt lgkmcnt(0)")
        e(f"v_add_f64 {v2(F)}, {v2(F)}, {v2(TT)}")
    e(f"s_lshl_b32 s{SVA}, s{SBASE}, 3")                   # the group's first node: byte offset
    e(f"s_lshr_b32 s{SVA + 1}, s{SBASE}, 29")
    e(f"s_add_u32 s{SVA}, s{SVA}, %[vlo]")
    e(f"s_addc_u32 s{SVA + 1}, s{SVA + 1}, %[vhi]")
    e("s_mov_b64 exec, 0xff")                              # lanes 0..7: one node each
    e(f"global_store_dwordx2 %[mvoff], {v2(F)}, {s2(SVA)}")
    e("s_mov_b64 exec, -1")


def epilogue(e, degree, volume):
    e("s_set_gpr_idx_off")
    if BLOCK and SPL == 6:
        # (the group's indices live in window 0 here: what the block's last row has requested into it -- the
        # harmless window behind the run's end -- must have landed first)
        e("s_waitcnt lgkmcnt(0)")
    # Group-level running maximum (nodes of a group are visited in ascending flat index: strict >).
    # A whole group (all eight nodes inside the grid, the common case) takes the first node's z
    # as the starting maximum; a group cut by the grid's edge starts from (-inf, none) and skips
    # the nodes outside.
    partial = e.label("pg")
    merge = e.label("mg")
    if LDS_STATE:
        e(f"v_mov_b32 v{VC}, %[clo]")
        e(f"v_mov_b32 v{VC + 1}, %[chi]")
    e(f"s_cmp_lg_u32 s{SMASK}, 0xff")
    e(f"s_cbranch_scc1 {partial}")
    if not LAZY:
        e(f"v_mov_b32 v{KI}, 0x7fffffff")                      # "no index" (a literal and vcc cannot
    butterfly = volume and MARGINAL and BUTTERFLY
    for g in range(8):                                         # feed one instruction)
        epilogue_node(e, degree, volume, g, g == 0 if not LAZY else (True if g == 0 else None if g == 1 else False),
                      defer=butterfly)
    if butterfly:
        marginal_butterfly(e)
    e(f"s_branch {merge}")
    e(f"{partial}:")
    for k in range(SPL):
        e(f"v_mov_b32 v{GMAX + 2 * k}, 0")
        e(f"v_mov_b32 v{GMAX + 2 * k + 1}, 0xfff00000")        # -inf
        if not LAZY:
            e(f"v_mov_b32 v{GIDX + k}, 0x7fffffff")
        if LDS_STATE:
            e(f"v_mov_b32 v{GSUM + 2 * k}, 0")
            e(f"v_mov_b32 v{GSUM + 2 * k + 1}, 0")
    for g in range(8):
        skip = e.label("nd")
        e(f"s_bitcmp1_b32 s{SMASK}, {g}")
        if LAZY:
            # a node outside the grid: NaN, so that it never equals the group's maximum below
            inside = e.label("in")
            e(f"s_cbranch_scc1 {inside}")
            for k in range(SPL):
                e(f"v_mov_b32 v{ACC + AS * g + 2 * k + 1}, 0x7ff80000")
            e(f"s_branch {skip}")
            e(f"{inside}:")
        else:
            e(f"s_cbranch_scc0 {skip}")
        epilogue_node(e, degree, volume, g, False)
        e(f"{skip}:")
    e(f"{merge}:")
    done = e.label("dn")
    if LAZY:
        # does any lane's group maximum reach its running maximum?  (>=: groups are not visited in
        # ascending flat index, an equal value may have to hand over a lower index)
        # (the group maxima are raw stacks: z = stack * scale, the rounded product every kernel
        # compares, once per group and sample slot -- into the dead TT registers)
        for k in range(SPL):
            e(f"v_mul_f64 {v2(TT + 2 * k)}, {v2(GMAX + 2 * k)}, %[scale]")
        if BMAX:
            # (the test below on z + slack instead of z: a group that comes NEAR the running maximum takes the
            # branch too, leaves its maxima in the brick's row and falls into the per-slot test on z itself)
            brick_max(e, TT, done)
        else:
            e(f"v_cmp_ge_f64 {s2(ST)}, {v2(TT)}, {MAXR[0]}")
            for k in range(1, SPL):
                e(f"v_cmp_ge_f64 vcc, {v2(TT + 2 * k)}, {MAXR[k]}")
                e(f"s_or_b64 {s2(ST)}, {s2(ST)}, vcc")
            e(f"s_cmp_eq_u64 {s2(ST)}, 0")
            e(f"s_cbranch_scc1 {done}")
        # yes: the eight nodes' flat indices into the (dead) F registers, then per sample slot k
        # that reached: the lowest node whose z equals the group's maximum (descending, so the
        # lowest is written last; nodes outside the grid hold NaN); a maximum of -inf has no node
        # (as the strict '>' from a (-inf, none) start); then the merge into the running pair:
        # larger z, ties -> lower flat index
        e(f"v_mov_b32 v{KI}, 0x7fffffff")                      # "no index"
        for g in range(8):
            node_index(e, g, False)
            e(f"v_mov_b32 v{F + g}, s{SNODE}")
        for k in range(SPL):
            nxt = e.label("nk")
            g_ = v2(TT + 2 * k)
            e(f"v_cmp_ge_f64 vcc, {g_}, {MAXR[k]}")
            e(f"s_cbranch_vccz {nxt}")
            e(f"v_mov_b32 v{GIDX + k}, v{KI}")
            for g in range(7, -1, -1):
                e(f"v_mul_f64 {v2(P)}, {v2(ACC + AS * g + 2 * k)}, %[scale]")      # the node's z
                e(f"v_cmp_eq_f64 vcc, {v2(P)}, {g_}")
                e(f"v_cndmask_b32 v{GIDX + k}, v{GIDX + k}, v{F + g}, vcc")
            e(f"v_cmp_gt_f64 vcc, {g_}, {s2(SNEGINF)}")
            e(f"v_cndmask_b32 v{GIDX + k}, v{KI}, v{GIDX + k}, vcc")
            e(f"v_cmp_gt_f64 {s2(ST)}, {g_}, {MAXR[k]}")
            e(f"v_cmp_eq_f64 {s2(ST + 2)}, {g_}, {MAXR[k]}")
            e(f"v_cmp_lt_i32 vcc, v{GIDX + k}, {IDXR[k]}")
            e(f"s_and_b64 {s2(ST + 2)}, {s2(ST + 2)}, vcc")
            e(f"s_or_b64 vcc, {s2(ST)}, {s2(ST + 2)}")
            e(f"v_cndmask_b32 {IDXR[k]}, {IDXR[k]}, v{GIDX + k}, vcc")
            e(f"v_max_f64 {MAXR[k]}, {MAXR[k]}, {g_}")
            e(f"{nxt}:")
        e(f"{done}:")
        return
    if BMAX:
        far = e.label("bf")
        brick_max(e, GMAX, far)
        e(f"{far}:")
    # merge into the wave's running pair: larger z, ties -> lower flat index
    if LDS_STATE:
        # the wavefront's running state lives in LDS (5 chunks of 64 lanes x 16 bytes): maxima into
        # the dead F registers, sums into P, indices into TT
        for c, reg in enumerate((F, F + 4, P, P + 4, TT)):
            e(f"ds_read_b128 v[{reg}:{reg + 3}], %[state] offset:{c * STATE_CHUNK}")
        e("s_waitcnt lgkmcnt(0)")
    for k in range(SPL):
        g_ = v2(GMAX + 2 * k)
        e(f"v_cmp_gt_f64 {s2(ST)}, {g_}, {MAXR[k]}")
        e(f"v_cmp_eq_f64 {s2(ST + 2)}, {g_}, {MAXR[k]}")
        e(f"v_cmp_lt_i32 vcc, v{GIDX + k}, {IDXR[k]}")
        e(f"s_and_b64 {s2(ST + 2)}, {s2(ST + 2)}, vcc")
        e(f"s_or_b64 vcc, {s2(ST)}, {s2(ST + 2)}")
        e(f"v_cndmask_b32 {IDXR[k]}, {IDXR[k]}, v{GIDX + k}, vcc")
        e(f"v_max_f64 {MAXR[k]}, {MAXR[k]}, {g_}")
    if LDS_STATE:
        for k in range(SPL):
            e(f"v_add_f64 {v2(P + 2 * k)}, {v2(P + 2 * k)}, {SUMR[k]}")
        for c, reg in enumerate((F, F + 4, P, P + 4, TT)):
            e(f"ds_write_b128 %[state], v[{reg}:{reg + 3}] offset:{c * STATE_CHUNK}")
    e(f"{done}:")


def brick_max(e, zreg, skip):
    """tie_rule = 1 on wide tiles (qm_ties.hpp): the refinement stacks again, per sample, the bricks whose largest
    z lies within the slack of the sample's maximum -- it needs to know them.  A node within the slack of the
    FINAL maximum is within the slack of the wavefront's running maximum when its group is merged (the running
    one is never larger), so: the group's maxima z_k (one per sample slot; present in both flavours at the merge)
    are tested as z_k + 2 slack(z_k) >= running maximum (twice the refinement's slack 4e-16 + 2^-50 |z|: the sum
    is rounded, and the slack is taken at z_k instead of the final maximum); a group no lane of which passes
    branches to `skip` -- the lazy flavour's own branch to its arg-max recovery, slightly widened --, in the
    others the lanes that pass raise the brick's row in global memory with an atomic maximum (the eight wavefronts
    of a workgroup share a brick's samples; rows start at -inf: qm_engine.hip fills them; a lane's running
    maximum is reached ~ln(groups) times in a walk: a few million atomics per C3 step).  Two VALU instructions
    per sample slot and group into the dead F registers, no register of its own: the row's address is
    %[brow] + the lane's LDS address (48 lane + the window base, which %[brow] has had subtracted)."""
    for k in range(SPL):
        e(f"v_fma_f64 {v2(F + 2 * k)}, |{v2(zreg + 2 * k)}|, %[slkrel], {v2(zreg + 2 * k)}")
    for k in range(SPL):
        e(f"v_add_f64 {v2(F + 2 * k)}, {v2(F + 2 * k)}, %[slkabs]")
    e(f"v_cmp_ge_f64 {s2(ST)}, {v2(F)}, {MAXR[0]}")
    for k in range(1, SPL):
        e(f"v_cmp_ge_f64 vcc, {v2(F + 2 * k)}, {MAXR[k]}")
        e(f"s_or_b64 {s2(ST)}, {s2(ST)}, vcc")
    e(f"s_cmp_eq_u64 {s2(ST)}, 0")
    e(f"s_cbranch_scc1 {skip}")
    # (only the lanes that passed, slot by slot: a wavefront has 384 running maxima, SOME lane's is reached by most
    # groups of a walk's first half -- all lanes' atomics on every such group cost the C3 step 18 %)
    for k in range(SPL):
        nxt = e.label("bk")
        e(f"v_cmp_ge_f64 vcc, {v2(F + 2 * k)}, {MAXR[k]}")
        e(f"s_cbranch_vccz {nxt}")
        e("s_mov_b64 exec, vcc")
        e(f"global_atomic_max_f64 %[lane], {v2(zreg + 2 * k)}, %[brow] offset:{8 * k}")
        e("s_mov_b64 exec, -1")
        e(f"{nxt}:")


def body(degree, volume):
    e = Emitter()
    if not LDS_STATE:
        # leading coefficient into a VGPR pair (two different SGPR pairs cannot feed one VALU op)
        e(f"v_mov_b32 v{VC}, %[clo]")
        e(f"v_mov_b32 v{VC + 1}, %[chi]")
    e(f"s_mov_b32 s{STAB}, %[tablo]")
    e(f"s_mov_b32 s{STAB + 1}, %[tabhi]")
    # prologue: lead-in record (header of row 0) and row 0's record; window of row 0
    e(f"s_load_dwordx{NBUF} s[{BUF[1]}:{BUF[1] + NBUF - 1}], {s2(STAB)}, 0")
    e(f"s_load_dwordx{NBUF} s[{BUF[0]}:{BUF[0] + NBUF - 1}], {s2(STAB)}, {REC}")
    e(f"s_mov_b32 s{SOFF}, {REC}")
    e(f"s_mov_b32 s{SNEGINF}, 0")
    e(f"s_mov_b32 s{SNEGINF + 1}, 0xfff00000")
    e(f"s_mov_b32 s{SMAGIC}, 0")
    e(f"s_mov_b32 s{SMAGIC + 1}, 0x43380000")
    e(f"v_mov_b32 v{VZERO}, 0")
    if LAZY:
        e(f"v_mov_b32 v{VMAG}, 0")
        e(f"v_mov_b32 v{VMAG + 1}, 0x43380000")
    e(f"s_add_u32 s{SPF}, s{STAB}, {PF_AHEAD * REC}")
    e(f"s_addc_u32 s{SPF + 1}, s{STAB + 1}, 0")
    if BLOCK and STAGE_IN_LOOP:
        e(f"s_mov_b32 s{SG_AFTER}, 0")
        if not FAR:
            e(f"s_mov_b32 s{SG_ROW}, %[stgrow]")
            stage_record(e)
    e("s_waitcnt lgkmcnt(0)")
    issue_window(e, 0, BUF[1] + R_HDR)
    group = e.label("grp")
    pair = e.label("pair")
    nopair = e.label("np")
    if BLOCK or NEXT_RUN:
        # the first lines of this wavefront's NEXT run (its next brick; row blocks: the next brick's first
        # block), into L2: one fire-and-forget load, lane l touches byte 64 l -- 4 KB = 128 records.  (The
        # per-row prefetch runs PF_AHEAD records ahead INSIDE a run; a run's first records would otherwise
        # be fetched from HBM by the scalar loads that wait for them.)
        e(f"global_load_dword v{VPF}, %[nxoff], %[nxrun]")
        # ... and the row-window metadata (16 bytes per row) that the staging of the next brick (row blocks:
        # of the block AFTER the next) will read, 4 KB of it: the staging otherwise waits for HBM twice in a
        # row, once for the metadata and once for the samples they point at
        if BLOCK or NEXT_META:
            e(f"global_load_dword v{VPF}, %[nxoff], %[mdrun]")
    if BLOCK:
        carry = e.label("cy")
        e("s_bitcmp1_b32 %[flags], 0")                         # first block of the brick: zero
        e(f"s_cbranch_scc0 {carry}")
        for r in range(ACC, ACC + 8 * AS):
            e(f"v_mov_b32 v{r}, 0")
        e(f"{carry}:")
    global STAGING
    STAGING = BLOCK and STAGE_IN_LOOP and not FAR
    e(f"{group}:")
    row_iter(e, 0, True)
    row_iter(e, 1, False)
    e(f"s_sub_u32 s{SPAIRS}, %[npairs], 2")                    # borrow <=> the group has one pair
    e(f"s_cbranch_scc1 {nopair}")
    if STAGING:
        # pairs of rows WITH the staging step while this wavefront has rows of the next block left to stage
        # (five of a block's seventeen), then the plain loop: the step's bookkeeping -- its skip test, the
        # record load, the count of loads behind the last staging load -- costs a pair of rows 4 %
        plain, staged = e.label("pl"), e.label("st")
        e(f"s_cmp_lt_u32 s{SG_ROW}, %[stgn]")
        e(f"s_cbranch_scc0 {plain}")
        e(f"{staged}:")
        row_iter(e, 0, False)
        row_iter(e, 1, False)
        e(f"s_sub_u32 s{SPAIRS}, s{SPAIRS}, 1")
        e(f"s_cbranch_scc1 {nopair}")
        e(f"s_cmp_lt_u32 s{SG_ROW}, %[stgn]")
        e(f"s_cbranch_scc1 {staged}")
        e(f"{plain}:")
        e(f"s_add_u32 s{SG_AFTER}, s{SG_AFTER}, s{SPAIRS}")     # the plain loop: s[SPAIRS] + 1 more pairs, one
        e(f"s_add_u32 s{SG_AFTER}, s{SG_AFTER}, 1")            # prefetch each
        STAGING = False
    e(f"{pair}:")
    row_iter(e, 0, False)
    row_iter(e, 1, False)
    e(f"s_sub_u32 s{SPAIRS}, s{SPAIRS}, 1")
    e(f"s_cbranch_scc0 {pair}")
    e(f"{nopair}:")
    if BLOCK:
        more = e.label("mb")
        e("s_set_gpr_idx_off")
        if STAGE_IN_LOOP and not FAR:
            # rows of the next block that this block's pairs of rows did not cover (a short last block in
            # front of a full one)
            again, done = e.label("sa"), e.label("sd")
            e(f"{again}:")
            e(f"s_cmp_ge_u32 s{SG_ROW}, %[stgn]")
            e(f"s_cbranch_scc1 {done}")
            stage_record(e)
            e("s_waitcnt lgkmcnt(0)")
            stage_step(e)
            e(f"s_branch {again}")
            e(f"{done}:")
        e("s_bitcmp1_b32 %[flags], 1")                         # last block: exponentiate, reduce
        e(f"s_cbranch_scc0 {more}")
        epilogue(e, degree, volume)
        e(f"{more}:")
    else:
        epilogue(e, degree, volume)
        e("s_sub_u32 %[ng], %[ng], 1")
        e("s_cmp_lg_u32 %[ng], 0")
        e(f"s_cbranch_scc1 {group}")
    if BLOCK and not volume and PF_AHEAD:
        # Loads return in order: the block staged before this call (LDS-direct loads) is in LDS once
        # no more loads are outstanding than this call issued itself -- its stream prefetches, one
        # per row, which nobody needs to wait for (a full vmcnt(0) here cost a round trip to HBM
        # per block).  2 * npairs of them at least: wait down to the largest step below that.
        # (round 5: the count is kept by the loop -- one prefetch per PAIR of rows since PF_EVERY = 2, and the
        # staging loads the loop itself now issues for the next block reset it)
        end = e.label("wv")
        e("s_waitcnt lgkmcnt(0)")
        if STAGE_IN_LOOP:
            count, steps = SG_AFTER, (24, 16, 12, 8, 4)
        else:
            count, steps = SPAIRS, (32, 24, 16, 8)
            e(f"s_lshl_b32 s{SPAIRS}, %[npairs], {1 if PF_EVERY == 1 else 0}")
        for step in steps:
            nxt = e.label("ws")
            e(f"s_cmp_ge_u32 s{count}, {step}")
            e(f"s_cbranch_scc0 {nxt}")
            e(f"s_waitcnt vmcnt({step})")
            e(f"s_branch {end}")
            e(f"{nxt}:")
        e("s_waitcnt vmcnt(0)")
        e(f"{end}:")
    else:
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return e.lines


def emit(degree, volume, lds_state, far, lazy, block, name, spl=4, contig=False, marginal=False, bmax=False):
    configure(lds_state, far, lazy, block, spl, contig, marginal, bmax)
    lines = body(degree, volume)
    # the stream pointer lives in a hard SGPR pair (the halves of an s[lo:hi] operand cannot be
    # named in inline asm): it is handed over as two 32-bit scalars
    text = "\\n\\t".join(lines)
    ragged = volume and contig and not marginal
    print()
    print(f"// degree-{degree} 2^f{', marginalised map' if marginal else ', values stored' if volume else ''}"
          f"{', running state in LDS' if lds_state else ''}{', far plane' if far else ''}"
          f"{', arg-max recovered lazily' if lazy else ''}{', brick maxima into LDS' if bmax else ''}"
          f"{', ONE group, one block of its rows per call (flags: 1 = first block, 2 = last)' if block else ''}"
          f"{f', WIDE tile of {spl} samples per lane, contiguous row windows from even samples' if spl == 6 else f', TAIL tile of {spl} sample(s) per lane, contiguous row windows' if contig else ''}; "
          f"window of up to {WMAX} doubles; "
          f"hard VGPRs v{VB}..v{VEND - 1}, SGPRs s{SB}..s{SEND - 1}")
    print(f"__device__ __forceinline__ void {name}("
          + ("" if lds_state else f"double (&vmax)[{spl}], double (&vsum)[{spl}], int (&vidx)[{spl}],"))
    print("        const void *stream, " + ("unsigned flags, const void *next_run, unsigned next_off, const void *next_meta, "
                                            if block else "int ngroups, const void *next_run, unsigned next_off, const void *next_meta, ")
          + "int npairs, unsigned lane_addr, "
          + ("const ShiftStageNext &stage, " if block and STAGE_IN_LOOP and not far else "")
          + ("unsigned state_addr, " if lds_state else "")
          + ("unsigned lane_addr_b, " if far else "") + ("const void *brick_row, " if bmax else "") + "int nz, "
          f"int nynz, double scale, const double (&c)[{degree + 1}]"
          + (f", double *marg_tile, const double (&w)[{spl}], unsigned node_off, unsigned lane_x16, "
             f"unsigned lane_x32" if marginal else
             f", double *vol_tile, unsigned vol_stride_bytes, unsigned lane_bytes, "
             f"const unsigned long long (&store_lanes)[{spl}]" if ragged else
             ", double *vol_tile, unsigned vol_stride_bytes, unsigned lane_bytes, "
             "unsigned long long store_lanes" if volume else "")
          + ") {")
    print("    const unsigned long long sp = (unsigned long long)stream;")
    print("    const unsigned tablo = (unsigned)sp, tabhi = (unsigned)(sp >> 32);")
    if block and STAGE_IN_LOOP and not far:
        print("    const unsigned long long gp = (unsigned long long)stage.src;")
        print("    const unsigned stglo = (unsigned)gp, stghi = (unsigned)(gp >> 32);")
    print(f"    const unsigned long long cl = (unsigned long long)__double_as_longlong(c[{degree}]);")
    print("    const unsigned clo = (unsigned)cl, chi = (unsigned)(cl >> 32);")
    if marginal:
        print("    const unsigned long long vp = (unsigned long long)marg_tile;")
        print("    const unsigned vlo = (unsigned)vp, vhi = (unsigned)(vp >> 32);")
    elif volume:
        print("    const unsigned long long vp = (unsigned long long)vol_tile;")
        print("    const unsigned vlo = (unsigned)vp, vhi = (unsigned)(vp >> 32);")
        if not ragged:
            print("    const unsigned mlo = (unsigned)store_lanes, mhi = (unsigned)(store_lanes >> 32);")
    outs = []
    if not lds_state:
        outs += [f'[max{k}] "+v"(vmax[{k}])' for k in range(spl)]
        outs += [f'[sum{k}] "+v"(vsum[{k}])' for k in range(spl)]
        outs += [f'[idx{k}] "+v"(vidx[{k}])' for k in range(spl)]
    if not block:
        outs += ['[ng] "+s"(ngroups)']
    ins = ['[tablo] "s"(tablo)', '[tabhi] "s"(tabhi)', '[lane] "v"(lane_addr)',
           '[npairs] "s"(npairs)', '[nz] "s"(nz)', '[nynz] "s"(nynz)', '[scale] "s"(scale)',
           '[clo] "s"(clo)', '[chi] "s"(chi)']
    if lds_state:
        ins += ['[state] "v"(state_addr)']
    if far:
        ins += ['[laneb] "v"(lane_addr_b)']
    if bmax:
        # (slack: twice qm_ties.hpp's tie_slack -- 2^-49 |z| + 8e-16)
        ins += ['[brow] "s"(brick_row)', '[slkrel] "s"(0x1p-49)', '[slkabs] "s"(8.0e-16)']
    if block:
        ins += ['[flags] "s"(flags)']
    if block and STAGE_IN_LOOP and not far:
        ins += ['[stgmeta] "s"(stage.meta)', '[stgn] "s"(stage.rows)', '[stgrow] "s"(stage.first_row)',
                '[stgstride] "s"(stage.stride)', '[stglo] "s"(stglo)', '[stghi] "s"(stghi)',
                '[stgt8] "s"(stage.row_bytes)', '[stglds] "s"(stage.lds)', '[lane32] "v"(stage.lane32)']
    if block or (NEXT_RUN and NEXT_META):
        ins += ['[mdrun] "s"(next_meta)']
    if block or NEXT_RUN:
        ins += ['[nxrun] "s"(next_run)', '[nxoff] "v"(next_off)']
    if marginal:
        ins += ['[vlo] "s"(vlo)', '[vhi] "s"(vhi)']
        ins += [f'[w{k}] "v"(w[{k}])' for k in range(spl)]
        if BUTTERFLY:
            ins += ['[mb0] "s"(0xaaaaaaaaaaaaaaaaull)', '[mb1] "s"(0xccccccccccccccccull)',
                    '[mb2] "s"(0xf0f0f0f0f0f0f0f0ull)', '[mvoff] "v"(node_off)', '[x16] "v"(lane_x16)',
                    '[x32] "v"(lane_x32)']
    elif ragged:
        ins += ['[vlo] "s"(vlo)', '[vhi] "s"(vhi)', '[vstride] "s"(vol_stride_bytes)',
                '[voff] "v"(lane_bytes)']
        ins += [f'[m{k}] "s"(store_lanes[{k}])' for k in range(spl)]
    elif volume:
        ins += ['[vlo] "s"(vlo)', '[vhi] "s"(vhi)', '[vstride] "s"(vol_stride_bytes)',
                '[voff] "v"(lane_bytes)', '[mlo] "s"(mlo)', '[mhi] "s"(mhi)']
    ins += [f'[c{i}] "s"(c[{i}])' for i in range(degree)]
    clob = [f'"v{r}"' for r in range(VB, VEND)] + [f'"s{r}"' for r in range(SB, SEND)]
    clob += ['"vcc"', '"scc"', '"m0"', '"memory"']
    print(f'    asm volatile("{text}"')
    print(f'                 : {", ".join(outs)}')
    print(f'                 : {", ".join(ins)}')
    print(f'                 : {", ".join(clob)});')
    print("}")


OVERLAY = "none"         # (tools/dev/shift_overlay.py says here what it changed)


def build_info():
    """what this file was generated from: the constants and a digest of the generator's source (the library
    hands it out through qm_build_info(); bench.py prints it, a GPU test asserts it is the product's)"""
    import hashlib
    import pathlib
    consts = " ".join(f"{k}={int(globals()[k])}" for k in (
        "NQMAX", "NQMIN", "NQMIN_WIDE", "PF_AHEAD", "PF_EVERY", "NEXT_RUN", "NEXT_META", "STAGE_IN_LOOP",
        "PACKED_GROUPS", "PACKED_BLOCKS", "PACKED_SHIFT64", "BUTTERFLY", "VOLUME_DEGREE",
        "MARGINAL_DEGREE", "VB_BLOCK", "VB_WIDE", "VB_WIDE_BLOCK"))
    digest = hashlib.sha256(pathlib.Path(__file__).read_bytes()).hexdigest()[:16]
    return f"{consts}; generator={digest}; overlay={OVERLAY}"


def main():
    print("// GENERATED by gen_shift_asm.py -- do not edit.  See that file for the schedule and the")
    print("// stream format.")
    print(f'constexpr char kShiftGenInfo[] = "{build_info()}";')
    configure(False)
    print(f"constexpr int kShiftNqMax = {NQMAX};          // quads (4 samples) a register window holds")
    print(f"constexpr int kShiftNqMin = {NQMIN};          // quads fetched unconditionally")
    print(f"constexpr int kShiftPlane = {PLANE2};        // bytes from plane A to plane B (4-wave workgroups)")
    print(f"constexpr int kShiftPlane3 = {PLANE3};       // ... of the 12-wave workgroup")
    print(f"constexpr int kShiftPlane8 = {PLANE8};       // ... of the 8-wave workgroup (33-64 rows)")
    print(f"constexpr int kShiftStateChunk = {STATE_CHUNK};   // LDS running state: 5 chunks per wavefront")
    print(f"constexpr bool kShiftPackedGroups = {'true' if PACKED_GROUPS else 'false'};   // 32-byte records (register indices as bytes)")
    print(f"constexpr bool kShiftPackedBlocks = {'true' if PACKED_BLOCKS else 'false'};   // ... of the row-block loops")
    print(f"constexpr int kShiftBlockVgprs = {VB_BLOCK};   // row-block flavour: the compiler's own code stays below")
    print("constexpr int kShiftWideSpl = 6;          // samples per lane of the wide tiles (time tile 384)")
    print(f"constexpr int kShiftWideBlockVgprs = {VB_WIDE_BLOCK};   // wide row-block flavour: the compiler's own code stays below")
    print(f"constexpr int kShiftNqMinWide = {NQMIN_WIDE};      // quads a wide tile's window fetches unconditionally")
    print(f"constexpr bool kShiftStageInLoop = {'true' if STAGE_IN_LOOP else 'false'};   // row blocks: the next block's staging issued by the row loop")
    print(f"constexpr int kShiftVolumeDegree = {VOLUME_DEGREE};   // 2^f polynomial of the volume-writing flavours")
    print(f"constexpr int kShiftMarginalDegree = {MARGINAL_DEGREE};   // 2^f polynomial of the marginal-map flavours")
    for degree, volume, lds_state, far, lazy, block, name in (
            (8, False, False, False, False, False, "shift_groups_detect"),
            (8, False, False, False, True, False, "shift_groups_detect_lazy"),
            (VOLUME_DEGREE, True, False, False, False, False, "shift_groups_volume"),
            (8, False, True, False, False, False, "shift_groups_detect3"),
            (8, False, False, True, False, False, "shift_groups_detect8"),
            (8, False, False, True, True, False, "shift_groups_detect8_lazy"),
            (VOLUME_DEGREE, True, False, True, False, False, "shift_groups_volume8"),
            (8, False, False, True, False, True, "shift_group_rows8"),
            (8, False, False, False, False, True, "shift_group_rows"),
            (8, False, False, False, True, True, "shift_group_rows_lazy"),
            (VOLUME_DEGREE, True, False, False, False, True, "shift_group_rows_volume")):
        emit(degree, volume, lds_state, far, lazy, block, name)
    # round 4: the marginalised map on full tiles (both workgroup shapes) ...
    emit(MARGINAL_DEGREE, True, False, False, False, False, "shift_groups_marginal", marginal=True)
    emit(MARGINAL_DEGREE, True, False, True, False, False, "shift_groups_marginal8", marginal=True)
    # ... and the tail tiles: 1, 2, 3 samples per lane, contiguous row windows (either shape)
    for spl in (1, 2, 3):
        emit(8, False, False, False, False, False, f"shift_tail{spl}_detect", spl, True)
        emit(VOLUME_DEGREE, True, False, False, False, False, f"shift_tail{spl}_volume", spl, True)
        emit(MARGINAL_DEGREE, True, False, False, False, False, f"shift_tail{spl}_marginal", spl, True, True)
    # round 6: WIDE tiles -- six samples per lane (time tile 384), contiguous row windows that start at an even
    # sample: 48 adds per register window instead of 32, 0.19 instead of 0.28 LDS reads per add at C3
    emit(8, False, False, False, False, False, "shift_wide_detect", 6, True)
    emit(8, False, False, False, True, False, "shift_wide_detect_lazy", 6, True)
    # ... with the brick maxima tie_rule = 1 refines from (brick_max)
    emit(8, False, False, False, False, False, "shift_wide_detect_bmax", 6, True, bmax=True)
    emit(8, False, False, False, True, False, "shift_wide_detect_bmax_lazy", 6, True, bmax=True)
    # ... and on ROW BLOCKS (tables whose 384-sample windows do not fit a CU's LDS all at once: 33 rows and up):
    # one group per wavefront, blocks of <= 20 rows through a double-buffered LDS, staged by the loop itself
    emit(8, False, False, False, False, True, "shift_wide_rows", 6, True)
    emit(8, False, False, False, True, True, "shift_wide_rows_lazy", 6, True)
    # (a volume-writing wide flavour -- three 16-byte stores per node at a lane stride of 48 bytes -- was built,
    # bit-equal, and measured: the whole 6000-sample C3 volume in 110 ms against 62 on the 256-sample tiles,
    # whose two stores at a stride of 32 bytes complete a 64-byte line from two lanes; here a store instruction
    # touches 48 lines to write 16: 1.5 x the write requests per sample.  Volume launches keep the 256-sample
    # tiles; profiles/r06_ab_runs.txt)


if __name__ == "__main__":
    main()
