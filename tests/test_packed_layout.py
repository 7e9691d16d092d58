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
    """ds_gemm_f16x2_plan: the row partition of the balanced launch (128x128 tiles + 64x64 tail tiles) -- main tiles
    fill whole rounds of the balance unit, the tail tiles cover exactly the remaining rows, the split lies on a packed
    row group, and the denoiser's shapes get the documented grids."""
    import ctypes as C
    from text_to_sound_synthesis_amd import _lib as L
    lib = L.lib()
    BM, BN, tbm, tbn = 128, 128, 64, 64

    def plan(M, N, store=L.STORE_ROW):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        L.check(lib.ds_gemm_f16x2_plan(0, M, N, store, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    ceil = lambda a, b: (a + b - 1) // b
    try:
        for slots in (512, 256, 64, 8, 1):
            lib.ds_gemm_f16x2_set_balance_slots(slots)
            for M in (1, 63, 128, 265, 530, 795, 2120, 4240, 16384, 16960, 33920):
                for N in (96, 256, 1024, 3072, 4096):
                    m_off, nbig, nsmall = plan(M, N)
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
                    assert plan(M, N, L.STORE_BATCH_T)[2] == 0          # transposed store: never split
        lib.ds_gemm_f16x2_set_balance_slots(512)
        # B = 64 (M = 16960) without the per-sample program: 2 / 6 / 8 rounds of 512 slots + 64x64 tiles on the last 576 rows
        assert plan(16960, 1024) == (16384, 1024, 9 * 16) and plan(16960, 3072) == (16384, 3072, 9 * 48)
        assert plan(16960, 4096) == (16384, 4096, 9 * 64)
    finally:
        lib.ds_gemm_f16x2_set_balance_slots(512)


def test_vt_store_unit_bookkeeping_mirror():
    """Host mirror of the V^T branch of the staged attention store (csrc/gemm_f16x2.hip: 8 consecutive keys of one d
    per 16-byte store, units aligned in a sample's own key index, a slab of SR rows touching at most two samples):
    every (sample, key) of the slab's valid rows is written exactly once from the right staged row, for the slab
    geometries of the 4-wave tiles (SR = 64 / 128; 256-row tiles in slabs stay covered as a property of the
    bookkeeping), including slabs that end or start past the last row."""
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


def test_per_sample_vt_store_mirror():
    """Host mirror of the V^T branch of the per-sample program's attention store (csrc/gemm_f16x2_ps.hip
    PS_SPLIT_SLAB): tile rows [m0, m0 + 288), m0 = 16 floor(265 b / 16), valid rows [off, off + 265), three slabs
    (tile rows 0..127, 128..255, 256..287); units of 8 keys aligned in the sample's own key index, a unit that
    straddles a slab edge written in two parts.  Every key of every sample is written exactly once, from the staged
    row that holds it, and nothing else is written."""
    L = 265
    for B in (1, 2, 3, 7, 16, 64):
        M = B * L
        for b in range(B):
            row_lo = b * L
            m0 = (row_lo >> 4) << 4
            off = row_lo - m0
            vhi = min(off + L, M - m0)
            assert 0 <= off <= 15 and off + L <= 288
            seen = {}
            for SL, SR in ((0, 128), (1, 128), (2, 32)):
                lo, hi = max(SL * 128, off), min(SL * 128 + SR, vhi)
                u_first = (lo - off) >> 3 if lo < hi else 0
                units = ((hi - off + 7) >> 3) - u_first if lo < hi else 0
                rlo, rhi = lo - SL * 128, hi - SL * 128
                for u in range(units):
                    k0 = (u_first + u) * 8
                    r0 = k0 + off - SL * 128
                    for e in range(8):
                        if rlo <= r0 + e < rhi:
                            assert 0 <= r0 + e < SR
                            key = k0 + e
                            assert key not in seen
                            seen[key] = m0 + SL * 128 + r0 + e          # global row that supplies this key
            assert seen == {k: row_lo + k for k in range(L)}, (B, b)


def test_ping_pong_schedule_hazards():
    """Happens-before check of the per-sample ping-pong main loop's schedule (csrc/gemm_f16x2_ps.hip).
    Model: both wave rows run the same phase program {ds_read; issue quarter g + LEAD; counted wait; barrier; MFMAs;
    barrier}, the second row one barrier behind; an event before barrier instance k (in either row) happens before every
    event after instance k (in both rows).  A wait in phase w retires this wave's quarters <= w + 2; a ds_read issued in
    phase g has completed by that row's second barrier of phase g.
      RAW: quarter q must be retired by BOTH rows' waits before any row reads it.
      WAR: a quarter may be issued on a region only after BOTH rows have completed the reads of its previous occupant,
           and never on a region read in the issuing phase.
    Quarter order {A-sub0, B-sub0, B-sub1, A-sub1 + ninth block}; B-sub0 is read in phase 0 (and stays in registers),
    the ninth block's rows (part of quarter type 3) in phase 3: LEAD 5 and 6 are legal, 7 is not.
    The counted wait: four consecutive quarters are one of each type = 2 + 2 + 2 + 3 instructions per wave = the
    vmcnt(9) of the steady state and of the prologue (quarters 0..5 issued, 0 and 1 needed)."""
    read_off = (0, 0, 1, 3)                # quarter type -> phase of its LAST read relative to its k-tile's phase 0
    first_off = (0, 0, 1, 2)               # ... and of its first read

    def b1(row, g):                        # barrier instance numbers (prologue barrier = 1, the extra one of row 1 = 2)
        return 2 * g + 2 + row

    def b2(row, g):
        return 2 * g + 3 + row

    def before_read(row, g):               # the barrier instance that precedes the ds_reads of phase g
        return b2(row, g - 1) if g > 0 else 1 + row

    def hazards(LEAD, nk=6):
        bad = []
        for q in range(4 * nk):
            tile, ty = divmod(q, 4)
            g_read = 4 * tile + first_off[ty]
            w = max(q - 2, -1)                                   # phase whose wait retires q (-1: the prologue wait)
            for reader in (0, 1):
                for waiter in (0, 1):
                    retired_at = b1(waiter, w) if w >= 0 else 1
                    if retired_at > before_read(reader, max(g_read, 0)):
                        bad.append(("RAW", q, reader, waiter))
            g_issue = q - LEAD
            if g_issue < 0 or tile < 2:
                continue                                         # prologue quarters / first use of a buffer
            g_prev = 4 * (tile - 2) + read_off[ty]               # last read of the region's previous occupant
            for reader in (0, 1):
                for issuer in (0, 1):
                    if b2(reader, g_prev) > before_read(issuer, g_issue):      # issue sits where the phase's reads sit
                        bad.append(("WAR", q, reader, issuer))
            if g_prev == g_issue:
                bad.append(("WAR-own-phase", q))
        return bad

    assert hazards(5) == [] and hazards(6) == []
    assert any(h[0] == "WAR" for h in hazards(7))
    size = (2, 2, 2, 3)                    # DMA instructions per wave of each quarter type
    for g in range(0, 40):                 # steady state: after issuing quarter g + 6 the wait leaves quarters g+3 .. g+6
        assert sum(size[(g + 6 - i) & 3] for i in range(4)) == 9
    assert sum(size[q & 3] for q in range(6)) == 13 and sum(size[q & 3] for q in (2, 3, 4, 5)) == 9   # prologue


def test_per_sample_tile_data_path_emulation():
    """Index plumbing of the per-sample 288 x 256 program (csrc/gemm_f16x2_ps.hip), emulated in numpy for the tiles of
    three samples over two k-tiles: packed planes in 'global memory' -> each wave's 16-row groups of the four quarters
    (incl. the ninth block's 4 groups riding with quarter type 3, loaded twice: waves w and w + 4) -> the 68 KB stage
    image -> the ds_read_b128 fragment addresses with the XOR chunk swizzle -> the operand layout of
    v_mfma_f32_32x32x16_f16 -> the accumulator layout, blocks 0..7 by (wave row, wave column) and block 8 by columns
    (wave (wr, wc) -> columns (2 wc + wr) 32) -> stores masked to the sample's rows.  The result must equal
    a1 b0 + a0 b1 + a0 b0 on every row of every sample, every output element written exactly once."""
    import numpy as np
    from text_to_sound_synthesis_amd import _lib as L
    rng = np.random.default_rng(3)
    Ls, B = 265, 3
    M, N, K, nk = B * Ls, 256, 64, 2
    Ah = rng.integers(-8, 9, size=(2, M, K)).astype(np.float16)           # [plane][row][k], small ints: exact sums
    Wh = rng.integers(-8, 9, size=(2, N, K)).astype(np.float16)
    Ap = L.pack_planes(torch.from_numpy(Ah)).numpy().reshape(2, -1)       # packed planes, flat halves per plane
    Wp = L.pack_planes(torch.from_numpy(Wh)).numpy().reshape(2, -1)
    BM, BN, HLD = 288, 256, 32
    APL, BPL = BM * HLD, BN * HLD
    STAGE = 2 * (APL + BPL)                                               # halves per buffer (68 KB)
    rgsA = (M + 15) >> 4
    C = np.zeros((M, N))
    done = np.zeros((M, N), dtype=np.int32)
    lane = np.arange(64)
    l31, hh = lane & 31, lane >> 5

    def mat(f):
        m = np.zeros((32, 16))
        for l in range(64):
            m[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = f[l]
        return m
    for b in range(B):
        row_lo = b * Ls
        m0, n0 = (row_lo >> 4) << 4, 0
        off = row_lo - m0
        vhi = min(off + Ls, M - m0)
        acc = np.zeros((BM, BN))
        for t in range(nk):
            lds = np.full(STAGE, np.nan, dtype=np.float32)
            written = np.zeros(STAGE, dtype=np.int32)
            for ty in range(4):
                for wave in range(8):
                    for k in range(2):
                        idx = 2 * wave + k
                        plane, r = idx >> 3, idx & 7
                        isA = ty in (0, 3)
                        sub = (ty == 3) if isA else (ty == 2)
                        gip = (r >> 2) * 8 + sub * 4 + (r & 3) if isA else (r >> 1) * 4 + sub * 2 + (r & 1)
                        rg = min(((m0 if isA else n0) >> 4) + gip, (rgsA if isA else N // 16) - 1)
                        src = (Ap if isA else Wp)[plane]
                        g_off = rg * nk * 512 + t * 512
                        l_off = (plane * 18 + gip if isA else 36 + plane * 16 + gip) * 512
                        lds[l_off:l_off + 512] = src[g_off:g_off + 512]
                        written[l_off:l_off + 512] += 1
                    if ty == 3:                                           # the ninth block's rows: piece e = wave & 3
                        e = wave & 3
                        plane, gip = e >> 1, 16 + (e & 1)
                        rg = min((m0 >> 4) + gip, rgsA - 1)
                        g_off = rg * nk * 512 + t * 512
                        l_off = (plane * 18 + gip) * 512
                        assert written[l_off] == (wave >> 2)              # waves 4..7 repeat waves 0..3: same bytes
                        lds[l_off:l_off + 512] = Ap[plane][g_off:g_off + 512]
                        written[l_off:l_off + 512] += 1
            w2 = written.reshape(-1, 512)[:, 0]
            assert (np.delete(w2, [16, 17, 34, 35]) == 1).all() and (w2[[16, 17, 34, 35]] == 2).all()
            for wave in range(8):
                wr, wc = wave >> 2, wave & 3
                for ks in range(2):
                    swz = ((2 * ks + hh) ^ ((l31 >> 2) & 3)) * 8
                    frag = lambda base_row, plane_off: mat(np.stack(
                        [lds[plane_off + (base_row + l31) * HLD + swz + e] for e in range(8)], axis=1))
                    bfr = {sb: (frag(wc * 64 + sb * 32, 2 * APL), frag(wc * 64 + sb * 32, 2 * APL + BPL)) for sb in (0, 1)}
                    for i in range(4):
                        a0, a1 = frag(wr * 128 + i * 32, 0), frag(wr * 128 + i * 32, APL)
                        for sb in (0, 1):
                            b0, b1 = bfr[sb]
                            acc[wr * 128 + i * 32:wr * 128 + i * 32 + 32, wc * 64 + sb * 32:wc * 64 + sb * 32 + 32] += \
                                a1 @ b0.T + a0 @ b1.T + a0 @ b0.T
                    e0, e1 = frag(256, 0), frag(256, APL)
                    b0, b1 = bfr[wr]
                    c8 = (2 * wc + wr) * 32
                    acc[256:288, c8:c8 + 32] += e1 @ b0.T + e0 @ b1.T + e0 @ b0.T
        for trow in range(off, vhi):
            C[m0 + trow] += acc[trow]
            done[m0 + trow] += 1
    assert (done == 1).all()
    A0, A1, W0, W1 = (x.astype(np.float64) for x in (Ah[0], Ah[1], Wh[0], Wh[1]))
    assert np.array_equal(C, A1 @ W0.T + A0 @ W1.T + A0 @ W0.T)


def test_ping_pong_schedule_hazards_16_row_ninth_block():
    """The same happens-before model for the NB16 variant of the per-sample program (272-row samples): the wave's B
    fragments for the ninth block row are read in PHASE 1 (so quarter type 1, B-sub0, is last read in phase 1 instead of 0)
    and the sixteen ninth-block A rows in PHASE 2 (type 3 is last read in phase 2 instead of 3).  LEAD 6 stays legal."""
    def hazards(LEAD, read_off, first_off, nk=6):
        b1 = lambda row, g: 2 * g + 2 + row
        b2 = lambda row, g: 2 * g + 3 + row
        before_read = lambda row, g: b2(row, g - 1) if g > 0 else 1 + row
        bad = []
        for q in range(4 * nk):
            tile, ty = divmod(q, 4)
            g_read = 4 * tile + first_off[ty]
            w = max(q - 2, -1)
            for reader in (0, 1):
                for waiter in (0, 1):
                    if (b1(waiter, w) if w >= 0 else 1) > before_read(reader, max(g_read, 0)):
                        bad.append(("RAW", q, reader, waiter))
            g_issue = q - LEAD
            if g_issue < 0 or tile < 2:
                continue
            g_prev = 4 * (tile - 2) + read_off[ty]
            for reader in (0, 1):
                for issuer in (0, 1):
                    if b2(reader, g_prev) > before_read(issuer, g_issue):
                        bad.append(("WAR", q, reader, issuer))
            if g_prev == g_issue:
                bad.append(("WAR-own-phase", q))
        return bad
    assert hazards(6, (0, 1, 1, 2), (0, 0, 1, 2)) == []
    # reading the B fragments as late as the 32-row variant reads its ninth block (phase 3) would race with the re-staging
    assert any(h[0].startswith("WAR") for h in hazards(6, (0, 3, 3, 3), (0, 0, 1, 2)))


def test_per_sample_tile_16_row_ninth_block_emulation():
    """Index plumbing of the NB16 variant (csrc/gemm_f16x2_ps.hip, samples of 272 rows = 17 packed groups): blocks 0..7 as
    in the 32-row variant; the ninth block row = rows 256..271, two 16 x 16 tiles per wave at columns (2 wc + wr) 32 + 16 tt
    on v_mfma_f32_16x16x32_f16 -- operand lane l holds row / column l & 15 and the 8 k values 8 (l >> 4) .., read from the
    stage image at chunk (l >> 4) ^ ((l & 15) >> 2 & 3); result lane l holds column l & 15, rows 4 (l >> 4) + r."""
    import numpy as np
    from text_to_sound_synthesis_amd import _lib as L
    rng = np.random.default_rng(5)
    Ls, B = 272, 2
    M, N, K, nk = B * Ls, 256, 64, 2
    Ah = rng.integers(-8, 9, size=(2, M, K)).astype(np.float16)
    Wh = rng.integers(-8, 9, size=(2, N, K)).astype(np.float16)
    Ap = L.pack_planes(torch.from_numpy(Ah)).numpy().reshape(2, -1)
    Wp = L.pack_planes(torch.from_numpy(Wh)).numpy().reshape(2, -1)
    BM, BN, HLD = 288, 256, 32
    APL, BPL = BM * HLD, BN * HLD
    STAGE = 2 * (APL + BPL)
    rgsA = (M + 15) >> 4
    C = np.zeros((M, N))
    done = np.zeros((M, N), dtype=np.int32)
    lane = np.arange(64)
    l31, hh, l15, kq = lane & 31, lane >> 5, lane & 15, lane >> 4

    def mat32(f):                      # 32x32x16 operand: lane l -> row l & 31, k = 8 (l >> 5) + e
        m = np.zeros((32, 16))
        for l in range(64):
            m[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = f[l]
        return m

    def mat16(f):                      # 16x16x32 operand: lane l -> row l & 15, k = 8 (l >> 4) + e
        m = np.zeros((16, 32))
        for l in range(64):
            m[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = f[l]
        return m
    for b in range(B):
        m0 = b * Ls
        assert m0 % 16 == 0
        acc = np.zeros((BM, BN))
        for t in range(nk):
            lds = np.full(STAGE, np.nan, dtype=np.float32)
            for ty in range(4):
                for wave in range(8):
                    for k in range(2):
                        idx = 2 * wave + k
                        plane, r = idx >> 3, idx & 7
                        isA = ty in (0, 3)
                        sub = (ty == 3) if isA else (ty == 2)
                        gip = (r >> 2) * 8 + sub * 4 + (r & 3) if isA else (r >> 1) * 4 + sub * 2 + (r & 1)
                        rg = min(((m0 if isA else 0) >> 4) + gip, (rgsA if isA else N // 16) - 1)
                        src = (Ap if isA else Wp)[plane]
                        l_off = (plane * 18 + gip if isA else 36 + plane * 16 + gip) * 512
                        lds[l_off:l_off + 512] = src[rg * nk * 512 + t * 512:rg * nk * 512 + t * 512 + 512]
                    if ty == 3:
                        e = wave & 3
                        plane, gip = e >> 1, 16 + (e & 1)
                        rg = min((m0 >> 4) + gip, rgsA - 1)
                        l_off = (plane * 18 + gip) * 512
                        lds[l_off:l_off + 512] = Ap[plane][rg * nk * 512 + t * 512:rg * nk * 512 + t * 512 + 512]
            for wave in range(8):
                wr, wc = wave >> 2, wave & 3
                for ks in range(2):
                    swz = ((2 * ks + hh) ^ ((l31 >> 2) & 3)) * 8
                    frag = lambda base_row, plane_off: mat32(np.stack(
                        [lds[plane_off + (base_row + l31) * HLD + swz + e] for e in range(8)], axis=1))
                    for i in range(4):
                        a0, a1 = frag(wr * 128 + i * 32, 0), frag(wr * 128 + i * 32, APL)
                        for sb in (0, 1):
                            b0, b1 = frag(wc * 64 + sb * 32, 2 * APL), frag(wc * 64 + sb * 32, 2 * APL + BPL)
                            acc[wr * 128 + i * 32:wr * 128 + i * 32 + 32, wc * 64 + sb * 32:wc * 64 + sb * 32 + 32] += \
                                a1 @ b0.T + a0 @ b1.T + a0 @ b0.T
                swzq = (kq ^ ((l15 >> 2) & 3)) * 8
                f16 = lambda base_row, plane_off: mat16(np.stack(
                    [lds[plane_off + (base_row + l15) * HLD + swzq + e] for e in range(8)], axis=1))
                ea0, ea1 = f16(256, 0), f16(256, APL)
                for tt in range(2):
                    col = (2 * wc + wr) * 32 + tt * 16
                    eb0, eb1 = f16(col, 2 * APL), f16(col, 2 * APL + BPL)
                    acc[256:272, col:col + 16] += ea1 @ eb0.T + ea0 @ eb1.T + ea0 @ eb0.T      # one 32-k MFMA per product
        for trow in range(0, Ls):
            C[m0 + trow] += acc[trow]
            done[m0 + trow] += 1
    assert (done == 1).all()
    A0, A1, W0, W1 = (x.astype(np.float64) for x in (Ah[0], Ah[1], Wh[0], Wh[1]))
    assert np.array_equal(C, A1 @ W0.T + A0 @ W1.T + A0 @ W0.T)


def test_generic_tile_data_path_emulation():
    """The same emulation for the generic staging of ds_gemm_f16x2_body (AMODE 2: wave w owns the 16-row groups
    w + NW i of the stage image, one DMA instruction each) over every tile geometry the library instantiates."""
    import numpy as np
    from text_to_sound_synthesis_amd import _lib as L
    rng = np.random.default_rng(5)
    K, nk = 32, 1
    for BM, BN, WGM, WGN in ((128, 128, 2, 2), (128, 64, 2, 2), (64, 64, 2, 2)):
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


def test_conv_fragment_packed_weights():
    """_lib.pack_conv_weights: the host-side layout ds_conv3x3_f16x2 (9 taps) / ds_conv1d_k3_f16x2 (3) / ds_convt1d_f16x2 (2 per
    phase) load straight into MFMA B operands: [Cout/128][Cin/32][taps][2 planes][4 blocks of 32 output channels = the wave]
    [2 k-steps][64 lanes][8 halves], lane (hh = lane >> 5, l = lane & 31) of fragment (wave, ks) holding
    W[128 nt + 32 wave + l][tap][32 slab + 16 ks + 8 hh + e]."""
    import random
    import torch
    from text_to_sound_synthesis_amd import _lib
    Cout, Cin = 256, 96
    for taps in (9, 3, 2):
        w = torch.randn(Cout, taps * Cin)
        planes, sc = _lib.split_f16x2(w)
        q = _lib.pack_conv_weights(planes, Cout, Cin, taps)
        assert q.numel() == 2 * Cout * taps * Cin
        if taps == 9:
            assert torch.equal(q, _lib.pack_conv3x3_weights(planes, Cout, Cin))
        q = q.view(Cout // 128, Cin // 32, taps, 2, 4, 2, 64, 8)
        pl = planes.view(torch.int16).view(2, Cout, taps, Cin)
        rng = random.Random(taps)
        for _ in range(3000):
            nt, ns, tap, p_, wave, ks, lane, e = [rng.randrange(n) for n in (Cout // 128, Cin // 32, taps, 2, 4, 2, 64, 8)]
            hh, l = lane >> 5, lane & 31
            assert q[nt, ns, tap, p_, wave, ks, lane, e] == pl[p_, nt * 128 + wave * 32 + l, tap, ns * 32 + ks * 16 + hh * 8 + e]
        # the planes reconstruct the scaled weights (hi + lo = w * 2^s to fp32 rounding of the split)
        back = (planes.view(torch.float16)[0].float() + planes.view(torch.float16)[1].float()) * sc
        assert (back - w).abs().max() <= 2e-7 * w.abs().max()


def test_vocoder_args_yml_reader(tmp_path):
    """pipeline.read_vocoder_args: the three integers of a MelGAN args.yml (an argparse.Namespace dump,
    evaluation/generate_samples_batch.py:34-36) read as plain text -- nothing is unpickled."""
    from text_to_sound_synthesis_amd.pipeline import read_vocoder_args
    f = tmp_path / "args.yml"
    f.write_text("!!python/object:argparse.Namespace\nbatch_size: 16\ndata_path: /x/y\nn_mel_channels: 80\nngf: 48  # wider\n"
                 "n_residual_layers: 4\nsave_path: logs/vggsound\n")
    assert read_vocoder_args(str(f)) == (80, 48, 4)
    f.write_text("ngf: 32\n")
    assert read_vocoder_args(str(f)) == (80, 32, 3)          # missing fields keep the reference configuration's values


def test_pack_operand_forms_and_subrange_arithmetic_host_mirror():
    """The host mirror of ds_pack_operand (csrc/pack.hip; the GPU test compares the kernel to exactly these expressions): the ROW
    form is pack_planes(split(X)), the TRANSPOSED form pack_planes(split(X^T zero-padded to rows_pad)); and the sub-range
    arithmetic modeling/train.py's _pack_parts relies on -- part i of a fused weight lands (row form) n0 * K halves into each
    plane and (transposed form) in k-range [n0, n0 + N_i) of X^T [K][N] -- holds in the documented layout."""
    from text_to_sound_synthesis_amd import _lib as L

    def split(a):
        hi = a.clamp(-65504.0, 65504.0).half()
        return torch.stack((hi, (a - hi.float()).clamp(-65504.0, 65504.0).half()))

    K, Ns = 64, (32, 96, 64)
    g = torch.Generator().manual_seed(5)
    parts = [torch.randn(n, K, generator=g) for n in Ns]
    fused = torch.cat(parts)
    N = fused.shape[0]
    row = L.pack_planes(split(fused)).reshape(2, -1)                     # [2][N * K]
    tr = L.pack_planes(split(fused.t().contiguous())).reshape(2, -1)     # X^T [K][N]: [2][ceil16(K) * N]
    n0 = 0
    for w in parts:
        n = w.shape[0]
        # row form: the part's own packing, dropped n0 * K halves into each plane of the fused operand
        own = L.pack_planes(split(w)).reshape(2, -1)
        assert torch.equal(row[:, n0 * K:(n0 + n) * K], own)
        # transposed form: element (k, n0 + j) of X^T sits at ds_packed_off(k, n0 + j, N / 32)
        own_t = split(w.t().contiguous())                                # [2][K][n]
        for k, j in ((0, 0), (5, 9), (17, n - 1), (K - 1, n // 2)):
            col = n0 + j
            off = ((k // 16) * (N // 32) + col // 32) * 512 + (k % 16) * 32 + (((col // 8) % 4) ^ ((k // 4) % 4)) * 8 + col % 8
            assert tr[0, off] == own_t[0, k, j] and tr[1, off] == own_t[1, k, j]
        n0 += n
