"""Host side of the packed split-plane layout (include/diffsound_hip.h, ds_gemm_desc.a_split): pack / unpack and the
documented address formula.  CPU only."""
import torch


def test_pack_planes_round_trip_and_formula():
    from text_to_sound_synthesis_amd import _lib as L
    R, K = 37, 96
    x2 = torch.arange(2 * R * K, dtype=torch.float32).remainder(2039).half().view(2, R, K)
    xp = L.pack_planes(x2)
    assert xp.numel() == 2 * 48 * K
    assert torch.equal(L.unpack_planes(xp, R, K), x2)
    flat = xp.view(2, -1)
    for r, k in ((0, 0), (5, 9), (17, 40), (36, 95), (12, 31)):       # the address formula of diffsound_hip.h
        off = ((r // 16) * (K // 32) + k // 32) * 512 + (r % 16) * 32 + (((k // 8) % 4) ^ ((r // 4) % 4)) * 8 + k % 8
        assert flat[0, off] == x2[0, r, k] and flat[1, off] == x2[1, r, k]


def test_balanced_launch_row_partition():
    """ds_gemm_f16x2_plan: the row partition of the balanced launches (default 128x128 + 64x64 tail; the big-tile
    candidates + 128x128 tail) -- main tiles fill whole rounds of the balance unit, the tail tiles cover exactly the
    remaining rows, the split lies on a packed row group, and the denoiser's shapes get the documented grids."""
    import ctypes as C
    from text_to_sound_synthesis_amd import _lib as L
    lib = L.lib()
    geo = {0: (128, 128, 64, 64), 7: (128, 128, 64, 64), 3: (256, 256, 128, 128), 4: (256, 128, 128, 128),
           6: (128, 256, 128, 128), 8: (256, 256, 128, 128)}

    def plan(cfg, M, N, store=L.STORE_ROW):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        L.check(lib.ds_gemm_f16x2_plan(cfg, M, N, store, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    ceil = lambda a, b: (a + b - 1) // b
    try:
        for slots in (512, 256, 64, 8, 1):
            lib.ds_gemm_f16x2_set_balance_slots(slots)
            lib.ds_gemm_f16x2_set_big_slots(slots)
            for cfg, (BM, BN, tbm, tbn) in geo.items():
                for M in (1, 63, 128, 265, 530, 795, 2120, 4240, 16384, 16960, 33920):
                    for N in (96, 256, 1024, 3072, 4096):
                        m_off, nbig, nsmall = plan(cfg, M, N)
                        tn = ceil(N, BN)
                        assert 0 < m_off <= M and nbig == ceil(m_off, BM) * tn
                        if nsmall == 0:
                            assert m_off == M
                            # no admissible split: no whole number of rounds fits, or the rows divide exactly
                            rbs = [rb for rb in range(1, M // BM + 1) if (rb * tn) % slots == 0]
                            assert not rbs or max(rbs) * BM == M
                        else:
                            assert m_off % BM == 0 and m_off % 16 == 0 and m_off < M
                            assert (m_off // BM * tn) % slots == 0
                            assert all(((rb * tn) % slots) for rb in range(m_off // BM + 1, M // BM + 1))   # the largest
                            assert nsmall == ceil(M - m_off, tbm) * ceil(N, tbn)
                        assert plan(cfg, M, N, L.STORE_BATCH_T)[2] == 0          # transposed store: never split
        lib.ds_gemm_f16x2_set_balance_slots(512)
        lib.ds_gemm_f16x2_set_big_slots(256)
        # B = 64 (M = 16960): default = 2 / 6 / 8 rounds of 512 slots + 64x64 tiles on the last 576 rows
        assert plan(0, 16960, 1024) == (16384, 1024, 9 * 16) and plan(0, 16960, 3072) == (16384, 3072, 9 * 48)
        assert plan(0, 16960, 4096) == (16384, 4096, 9 * 64)
        # candidates: one workgroup per CU -> 1 / 3 / 4 rounds of 256x256 tiles, 128x128 tiles on the last 576 rows
        assert plan(3, 16960, 1024) == (16384, 256, 5 * 8) and plan(3, 16960, 3072) == (16384, 768, 5 * 24)
        assert plan(3, 16960, 4096) == (16384, 1024, 5 * 32) and plan(4, 16960, 1024) == (16384, 512, 5 * 8)
    finally:
        lib.ds_gemm_f16x2_set_balance_slots(512)
        lib.ds_gemm_f16x2_set_big_slots(256)


def test_vt_store_unit_bookkeeping_mirror():
    """Host mirror of the V^T branch of the staged attention store (csrc/gemm_f16x2.hip: 8 consecutive keys of one d
    per 16-byte store, units aligned in a sample's own key index, a slab of SR rows touching at most two samples):
    every (sample, key) of the slab's valid rows is written exactly once from the right staged row, for the slab
    geometries of the 4-wave tiles (SR = 64 / 128) and of the big-tile candidates (SR = 128 slabs of 256-row tiles,
    SR = 256), including slabs that end or start past the last row."""
    def slab_writes(ms, SR, M, L, row_off=0):
        g0 = ms + row_off
        b0, pos0 = divmod(g0, L)
        rows_here = min(M - ms, SR)
        if rows_here <= 0:
            return {}
        seg0 = min(L - pos0, rows_here)
        u0 = ((pos0 + seg0 + 7) >> 3) - (pos0 >> 3)
        seg1 = rows_here - seg0
        units = u0 + ((seg1 + 7) >> 3)
        out = {}
        for u in range(units):
            first = u < u0
            b = b0 if first else b0 + 1
            k0 = ((pos0 >> 3) + u) * 8 if first else (u - u0) * 8
            rl0 = k0 - pos0 if first else seg0 + k0
            lo_ok, hi_ok = (0, seg0) if first else (seg0, rows_here)
            for e in range(8):
                if lo_ok <= rl0 + e < hi_ok:
                    assert (b, k0 + e) not in out
                    out[(b, k0 + e)] = rl0 + e               # staged row (slab-local) that supplies this key
        return out

    L = 265
    for B in (1, 2, 3, 8):
        M = B * L
        for BM, SR in ((64, 64), (128, 128), (256, 128), (256, 256)):
            for row_off in (0, 128 * 3):
                seen = {}
                for m0 in range(0, M, BM):
                    for sl in range(BM // SR):
                        ms = m0 + sl * SR
                        assert SR <= L + 1                      # a slab never touches three samples
                        for (b, key), rl in slab_writes(ms, SR, M, L, row_off).items():
                            assert (b, key) not in seen
                            seen[(b, key)] = ms + rl + row_off
                want = {divmod(r + row_off, L): r + row_off for r in range(M)}
                assert seen == want, (B, BM, SR, row_off)


def test_ping_pong_schedule_hazards():
    """Happens-before check of the ping-pong main loop's schedule (csrc/gemm_f16x2.hip AMODE 4, probe pp_kernel).
    Model: both wave rows run the same phase program {ds_read; issue quarter g + LEAD; counted wait; barrier; MFMAs;
    barrier}, the second row one barrier behind; an event before barrier instance k (in either row) happens before every
    event after instance k (in both rows).  A wait in phase w retires this wave's quarters <= w + 2; a ds_read issued in
    phase g has completed by that row's second barrier of phase g.
      RAW: quarter q must be retired by BOTH rows' waits before any row reads it.
      WAR: a quarter may be issued on a region only after BOTH rows have completed the reads of its previous occupant,
           and never on a region read in the issuing phase.
    Order {A-sub0, B-sub0, B-sub1, A-sub1} (B-sub0 kept in registers): LEAD 5 and 6 are legal, 7 is not.
    Order {B-sub0, A-sub0, B-sub1, A-sub1} with B-sub0 read one phase early (the probe's FLAGS bit 2): 5, 6, 7 legal, 8 not."""
    # per order: quarter type -> phase of its (only / last) read relative to its k-tile's phase 0
    orders = {"a-first": (0, 0, 1, 2), "b-first": (-1, 0, 1, 2)}

    def b1(row, g):                        # barrier instance numbers (prologue barrier = 1, the extra one of row 1 = 2)
        return 2 * g + 2 + row

    def b2(row, g):
        return 2 * g + 3 + row

    def before_read(row, g):               # the barrier instance that precedes the ds_reads of phase g
        return b2(row, g - 1) if g > 0 else 1 + row        # (g = -1, the prologue read of b-first: after the same)

    def hazards(read_off, LEAD, nk=6):
        bad = []
        for q in range(4 * nk):
            tile, ty = divmod(q, 4)
            g_read = 4 * tile + read_off[ty]
            w = max(q - 2, -1)                                   # phase whose wait retires q (-1: the prologue wait)
            for reader in (0, 1):
                for waiter in (0, 1):
                    retired_at = b1(waiter, w) if w >= 0 else 1
                    if retired_at > before_read(reader, max(g_read, 0)):
                        bad.append(("RAW", q, reader, waiter))
            g_issue = q - LEAD
            if g_issue < 0 or tile < 2:
                continue                                         # prologue quarters / first use of a buffer
            g_prev = 4 * (tile - 2) + read_off[ty]               # (last) read of the region's previous occupant
            for reader in (0, 1):
                for issuer in (0, 1):
                    if b2(reader, g_prev) > before_read(issuer, g_issue):      # issue sits where the phase's reads sit
                        bad.append(("WAR", q, reader, issuer))
            if g_prev == g_issue:
                bad.append(("WAR-own-phase", q))
        return bad

    assert hazards(orders["a-first"], 5) == [] and hazards(orders["a-first"], 6) == []
    assert any(h[0] == "WAR" for h in hazards(orders["a-first"], 7))
    for lead in (5, 6, 7):
        assert hazards(orders["b-first"], lead) == [], lead
    assert any(h[0] == "WAR" for h in hazards(orders["b-first"], 8))


def test_ping_pong_tile_data_path_emulation():
    """Index plumbing of the 256x256 ping-pong program (csrc/gemm_f16x2.hip AMODE 4 / probe pp_kernel), emulated in
    numpy for one tile over two k-tiles: packed planes in 'global memory' -> the wave's two 16-row groups of each
    16 KB quarter (q_src / q_lds) -> the 64 KB stage image -> the ds_read_b128 fragment addresses with the XOR chunk
    swizzle (per wave row / column, A-sub / B-sub) -> the operand layout of v_mfma_f32_32x32x16_f16 (lane l holds row
    l % 32, k = 8 (l / 32) .. +7) -> the accumulator layout (col = l % 32, row = (r & 3) + 8 (r >> 2) + 4 (l / 32)).
    The result must equal a1 b0 + a0 b1 + a0 b0 summed over k, and every byte of a stage must be written exactly once."""
    import numpy as np
    from text_to_sound_synthesis_amd import _lib as L
    rng = np.random.default_rng(3)
    BM = BN = 256
    K, nk = 64, 2
    Ah = rng.integers(-8, 9, size=(2, BM, K)).astype(np.float16)          # [plane][row][k], small ints: exact sums
    Wh = rng.integers(-8, 9, size=(2, BN, K)).astype(np.float16)
    Ap = L.pack_planes(torch.from_numpy(Ah)).numpy().reshape(2, -1)       # packed planes, flat halves per plane
    Wp = L.pack_planes(torch.from_numpy(Wh)).numpy().reshape(2, -1)
    HLD, APL, BPL = 32, BM * 32, BN * 32
    STAGE = 2 * (APL + BPL)                                               # halves per buffer
    m0 = n0 = 0
    for order in ("a-first", "b-first"):
        lds = np.full((2, STAGE), np.nan, dtype=np.float32)               # two buffers, in halves
        written = np.zeros((2, STAGE), dtype=np.int32)

        def issue(tile, ty, buf):
            for wave in range(8):
                for k in range(2):
                    idx = 2 * wave + k
                    plane, r = idx >> 3, idx & 7
                    isA = (ty & 1) == 1 if order == "b-first" else ty in (0, 3)
                    sub = (ty == 3) if isA else (ty == 2)
                    gip = (r >> 2) * 8 + sub * 4 + (r & 3) if isA else (r >> 1) * 4 + sub * 2 + (r & 1)
                    rg = ((m0 if isA else n0) >> 4) + gip
                    src = (Ap if isA else Wp)[plane]
                    g_off = rg * nk * 512 + tile * 512                   # halves: one packed 16-row x 32-k tile = 512
                    l_off = ((0 if isA else 32) + plane * 16 + gip) * 512    # 1 KB per group = 512 halves
                    lds[buf, l_off:l_off + 512] = src[g_off:g_off + 512]    # lane l moves 8 halves at 8 l: linear copy
                    written[buf, l_off:l_off + 512] += 1

        C = np.zeros((BM, BN), dtype=np.float64)
        lane = np.arange(64)
        l31, hh = lane & 31, lane >> 5
        for t in range(nk):
            buf = t & 1
            for ty in range(4):
                issue(t, ty, buf)
            assert (written[buf] == t // 2 + 1).all()                     # the four quarters tile the 64 KB stage
            for wave in range(8):
                wr, wc = wave >> 2, wave & 3
                for ks in range(2):
                    swz = ((2 * ks + hh) ^ ((l31 >> 2) & 3)) * 8          # halves inside the 32-half row
                    frag = lambda base_row, plane_off: np.stack(
                        [lds[buf, plane_off + (base_row + l31) * HLD + swz + e] for e in range(8)], axis=1)   # [lane][8]
                    for sa in range(2):
                        for ib in range(2):
                            arow = wr * 128 + sa * 64 + ib * 32
                            a0, a1 = frag(arow, 0), frag(arow, APL)
                            for sb in range(2):
                                bcol = wc * 64 + sb * 32
                                b0, b1 = frag(bcol, 2 * APL), frag(bcol, 2 * APL + BPL)
                                # MFMA 32x32x16: operand row = lane % 32, k slots 8 (lane / 32) .. +7
                                def mm(a, b):
                                    am = np.zeros((32, 16)); bm = np.zeros((32, 16))
                                    for l in range(64):
                                        am[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a[l]
                                        bm[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = b[l]
                                    return am @ bm.T                       # [row of A block][row of B block = column]
                                d = mm(a1, b0) + mm(a0, b1) + mm(a0, b0)
                                # accumulator register r of lane l: col = l % 32, row = (r & 3) + 8 (r >> 2) + 4 (l / 32);
                                # the epilogue maps block (i = 2 sa + ib, j = sb) to rows (wr 4 + i) 32.., cols (wc 2 + j) 32..
                                i, j = 2 * sa + ib, sb
                                for r in range(16):
                                    for l in range(64):
                                        row = (wr * 4 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                                        col = (wc * 2 + j) * 32 + (l & 31)
                                        # d is indexed [A row in block][B row in block]; the lane's register r holds
                                        # exactly that element
                                        C[row, col] += d[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
        A0, A1 = Ah[0].astype(np.float64), Ah[1].astype(np.float64)
        W0, W1 = Wh[0].astype(np.float64), Wh[1].astype(np.float64)
        want = A1 @ W0.T + A0 @ W1.T + A0 @ W0.T
        assert np.array_equal(C, want), order


def test_generic_tile_data_path_emulation():
    """The same emulation for the generic staging of ds_gemm_f16x2_body (AMODE 2: wave w owns the 16-row groups
    w + NW i of the stage image, one DMA instruction each) over every tile / wave-grid geometry the library
    instantiates -- the measured 4-wave tiles and the 8-wave candidates incl. the tail program."""
    import numpy as np
    from text_to_sound_synthesis_amd import _lib as L
    rng = np.random.default_rng(5)
    K, nk = 32, 1
    for BM, BN, WGM, WGN in ((128, 128, 2, 2), (128, 64, 2, 2), (64, 64, 2, 2),
                             (256, 256, 2, 4), (256, 128, 4, 2), (128, 256, 2, 4), (128, 128, 2, 4)):
        NW, TM, TN = WGM * WGN, BM // (32 * WGM), BN // (32 * WGN)
        Ah = rng.integers(-8, 9, size=(2, BM, K)).astype(np.float16)
        Wh = rng.integers(-8, 9, size=(2, BN, K)).astype(np.float16)
        Ap = L.pack_planes(torch.from_numpy(Ah)).numpy().reshape(2, -1)
        Wp = L.pack_planes(torch.from_numpy(Wh)).numpy().reshape(2, -1)
        APL, BPL = BM * 32, BN * 32
        STAGE = 2 * (APL + BPL)
        G = 2 * (BM + BN) // 16 // NW
        assert G * NW * 16 == 2 * (BM + BN)
        lds = np.full(STAGE, np.nan, dtype=np.float32)
        written = np.zeros(STAGE, dtype=np.int32)
        for wave in range(NW):
            for i in range(G):
                r = 16 * (wave + NW * i)                                 # first row of the group in the stage image
                if r < 2 * BM:
                    plane, rr, src = (1, r - BM, Ap) if r >= BM else (0, r, Ap)
                else:
                    r2 = r - 2 * BM
                    plane, rr, src = (1, r2 - BN, Wp) if r2 >= BN else (0, r2, Wp)
                g_off = (rr >> 4) * nk * 512
                l_off = (wave + NW * i) * 512                            # d_ = wave * 1024 + i * NW * 1024 bytes
                lds[l_off:l_off + 512] = src[plane][g_off:g_off + 512]
                written[l_off:l_off + 512] += 1
        assert (written == 1).all()
        C = np.zeros((BM, BN))
        done = np.zeros((BM, BN), dtype=np.int32)
        lane = np.arange(64)
        l31, hh = lane & 31, lane >> 5
        for wave in range(NW):
            wm, wn = wave // WGN, wave % WGN
            for ks in range(2):
                swz = ((2 * ks + hh) ^ ((l31 >> 2) & 3)) * 8
                frag = lambda base_row, plane_off: np.stack(
                    [lds[plane_off + (base_row + l31) * 32 + swz + e] for e in range(8)], axis=1)

                def mat(f):
                    m = np.zeros((32, 16))
                    for l in range(64):
                        m[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = f[l]
                    return m
                for i in range(TM):
                    a0, a1 = mat(frag((wm * TM + i) * 32, 0)), mat(frag((wm * TM + i) * 32, APL))
                    for j in range(TN):
                        b0, b1 = mat(frag((wn * TN + j) * 32, 2 * APL)), mat(frag((wn * TN + j) * 32, 2 * APL + BPL))
                        d = a1 @ b0.T + a0 @ b1.T + a0 @ b0.T
                        r0, c0 = (wm * TM + i) * 32, (wn * TN + j) * 32
                        C[r0:r0 + 32, c0:c0 + 32] += d
                        if ks == 0:
                            done[r0:r0 + 32, c0:c0 + 32] += 1
        assert (done == 1).all(), (BM, BN)
        A0, A1, W0, W1 = (x.astype(np.float64) for x in (Ah[0], Ah[1], Wh[0], Wh[1]))
        assert np.array_equal(C, A1 @ W0.T + A0 @ W1.T + A0 @ W0.T), (BM, BN, WGM, WGN)
