// The single-pass emitter (default): search, classify and emit in one kernel.  Included by kernels.hip after pairs.inl, inside
// namespace arp.  Replaces the reference's R*-tree walk + per-pair rules (src/contacts/complex.rs:189-299).
//
// Shape (as k_pairs, pairs.inl): task = 64 consecutive slots of the cell-sorted order, lane = home atom, five contiguous slot
// windows per lane (half shell; cells are kx times finer along x so the windows hug the search sphere), neighbour records staged
// through a wave-private LDS chunk, an f32 prefilter with a proven margin, survivors compacted into an LDS queue, exact f64
// phase on full waves of 64 survivors.
//
// What round 2's counters said (profiles/r02_pmc_summary.txt): the vector ALU is the pipe that binds (91 M instructions per
// launch on the headline input, two thirds of them in the exact phase, ~130 per batch of 64 survivors), the LDS pipe second
// (half of its cycles in the exact phase's table reads).  So this kernel is an instruction diet around the same data flow:
//   * the wave-uniform bounds (cutoff^2, 3.5 / 4.0 / 4.5 A) live in scalar registers, not in LDS;
//   * all rows of a pair come out of ONE 2048-entry LDS table indexed by {seven pair predicates, distance level against the
//     three fixed bounds, distance level against the element pair's three bounds} -- the levels are counted with
//     v_cmp_lt_f64 + v_addc_co_u32 (two instructions per bound), nothing is selected or branched on per rule;
//   * every lane computes every step (no exec-masked regions inside a batch: a masked region costs scalar bookkeeping and
//     saves no issue slot), only the stores are predicated;
//   * the chain key {res_ord, chain | model} of an atom is one aligned 64-bit word, so "which atom is the ligand" is one
//     64-bit compare (complex.rs:108-130 for chain groups "/");
//   * the prefilter runs in groups of 8 tests (lockstep granularity: a wave runs as long as its longest window) and
//     compacts once per 32 tests.
// Round 4 (profiles/r04_emit_experiments.txt, r04_issue_mix_microbench.txt): a scalar instruction is not free either -- it holds the SIMD's
// one scalar issue slot for four cycles, as long as a half-rate vector instruction -- and round 3's batch ran ~74 of them.  The scalar diet:
// masks straight out of VOP3 compares, one rare-path branch and a single exit per batch, the allocator's common case hand-scheduled, v_cmpx
// compaction rounds, prefilter runs on immediate offsets; the short level count; 24-bit slot entries (one multiply to the gather offset).
// What the time is made of now was measured by DOUBLING each class of work: the scattered gathers (21 us per million wave-loads), the LDS
// reads (5.2) and every vector (0.63) or scalar (0.77) instruction add up to the kernel's time -- no single pipe binds.
constexpr uint32_t kEChunk = 128;             // staged neighbour records per chunk
constexpr int kEWaves = 12;                   // waves per block: two blocks per CU share the CU's LDS, 6 waves per SIMD
constexpr uint32_t kEBlocks = 256u * 2u;
constexpr int kEWavesPerSimd = (kEWaves * 2) / 4;
constexpr uint32_t kESlotMask = (1u << kESlotBits) - 1u;
constexpr uint32_t kEGroup = 8;               // prefilter tests per lane between two "is any window still open" checks
constexpr uint32_t kEAcc = 32;                // prefilter tests per lane between two compactions (one mask word)
constexpr uint32_t kEQueue = 128;             // the queue fills to 63 + 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct WaveLdsE {
    float4 nrec[kEChunk + kEGroup];           // f32 prefilter records of the staged chunk (+ kEGroup: over-reads stay in bounds)
    u32x4 hxy[64]; u32x4 hzp[64];             // the task's home atoms: {x, y}, {z, pw, orig} as raw words ...
    unsigned long long hkey[64];              // ... and the chain key crm << 32 | res_ord
    uint32_t queue[kEQueue];                  // phase-1 survivors: home lane << 26 | neighbour slot
};
struct WaveLdsR { WaveLdsE e; uint32_t nkey[kEChunk + kEGroup]; };  // RES: + the residue words of the staged chunk (Sorted::rkey)
struct WaveLdsN { WaveLdsE e; };
struct alignas(16) TablesE {                  // block-shared decision tables (10 KB)
    double s_vdw[256];                        // van-der-Waals bound of the element pair (vdw.rs:41), index = class a << 4 | class b
    uint32_t lut[2048];                       // rows of a pair: index = W | Lg << 7 | Le << 9 (pair_lut2_entry)
    // (round 5: the clash and covalent bounds of the element pair, vdw.rs:33-34, are read from the parameter block in global memory: they are
    // only looked at behind m_close -- a candidate below the largest covalent bound of any element pair present -- and their 4 KB of LDS per
    // block are what the residue words of the RES kernels' staged chunks live in)
};

// Table entry for the seven pair predicates W (Fat::pw, classify_fast), the level Lg against the fixed bounds {4.5, 4.0, 3.5} and the
// level Le against the element pair's bounds {vdw, cov, clash} (both nested, so the level is a count).  Bits 0-18: the rows that need no
// probe (complex.rs:217-296), bit 30: a hydrogen probe decides if either residue carries hydrogens (hbond.rs:37,81), bit 29: the
// disulfide dihedral decides if residue tables were given (vdw.rs:46-53).  A steric clash ends the pair (complex.rs:233-235).
__host__ __device__ constexpr inline uint32_t pair_lut2_entry(uint32_t idx) {
    const uint32_t le = idx >> 9, base = pair_lut_entry(idx & 0x1FFu);
    if (le == 3u) return 1u << ARP_StericClash;
    uint32_t k = base & 0x1FFFFFFFu;
    k |= le == 2u ? (1u << ARP_CovalentBond) : (le == 1u ? (1u << ARP_VanDerWaalsContact) : 0u);
    k |= base & (1u << 30);
    k |= le == 2u ? (base & (1u << 29)) : 0u;
    return k;
}
// the table does not depend on the call: evaluated at compile time, it sits in the code object and a block copies it (8 KB out of L2)
struct Lut2 { uint32_t v[2048]; };
constexpr Lut2 make_lut2() {
    Lut2 t{};
    for (uint32_t k = 0; k < 2048u; k++) t.v[k] = pair_lut2_entry(k);
    return t;
}
__device__ const Lut2 kLut2 = make_lut2();
DEVFN void load_tables_e(TablesE &tb, const DevParams *dprm) {
    const double *src = dprm->s_vdw;
    double *dst = tb.s_vdw;
    for (uint32_t k = threadIdx.x; k < 256u; k += blockDim.x) dst[k] = src[k];
    const uint4 *lsrc = reinterpret_cast<const uint4 *>(kLut2.v);
    uint4 *ldst = reinterpret_cast<uint4 *>(tb.lut);
    for (uint32_t k = threadIdx.x; k < 512u; k += blockDim.x) ldst[k] = lsrc[k];
}

// Compaction rounds as one loop in assembly.  Every round, each lane with a surviving test (mask != 0) appends tag + (count of leading
// zeros of its mask) -- home lane << 26 | neighbour slot -- at the queue tail + 4 * (its rank among those lanes) and clears that bit;
// qbyte is the LDS byte address of the tail.  Returns when no lane has a test left or when the queue holds a full batch (qbyte >= qfull).
// Left to the compiler the same loop costs ~24 instructions per round (address rebuilt from an entry count, lane mask copied twice, loop
// condition recomputed); here 7 vector + 5 scalar + the LDS write + 2 branches.  A wave issues roughly one instruction per 5-9 cycles
// whatever its kind (tests/microbench/valu_rate.hip), and a task runs ~60 rounds.
DEVFN void compact_rounds_e(uint32_t &mask, uint32_t tag, uint32_t &qbyte, uint32_t qfull) {
    unsigned long long save;
    uint32_t t, lz, ent, bm, c;
    // Round 4: v_cmpx narrows exec itself, round after round (a lane whose mask has run empty stays out: its mask stays empty), so the loop
    // carries no exec save / restore -- 5 scalar instructions per round instead of 7, each a 4-cycle slot of the SIMD's scalar issue
    // (tests/microbench/issue_mix.hip).  exec is restored once, on the way out.
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "1:\n\t"
        "v_cmpx_ne_u32_e32 vcc, 0, %[mask]\n\t"
        "s_cbranch_vccz 2f\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_ffbh_u32 %[lz], %[mask]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[qb]\n\t"
        "v_add_u32 %[ent], %[tag], %[lz]\n\t"
        "v_lshrrev_b32 %[bm], %[lz], %[top]\n\t"
        "s_bcnt1_i32_b64 %[c], vcc\n\t"
        "ds_write_b32 %[t], %[ent]\n\t"
        "v_xor_b32 %[mask], %[mask], %[bm]\n\t"
        "s_lshl2_add_u32 %[qb], %[c], %[qb]\n\t"
        "s_cmp_lt_u32 %[qb], %[qf]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "2:\n\t"
        "s_mov_b64 exec, %[save]"
        : [mask] "+v"(mask), [qb] "+s"(qbyte), [t] "=&v"(t), [lz] "=&v"(lz), [ent] "=&v"(ent), [bm] "=&v"(bm), [c] "=&s"(c), [save] "=&s"(save)
        : [tag] "v"(tag), [top] "s"(0x80000000u), [qf] "s"(qfull)
        : "vcc", "scc", "memory");
}

// The same rounds with the reference's residue rule applied to every survivor BEFORE it is appended (k_emit<.., RES>): two atoms of one residue,
// or of sequence neighbours in one chain, never pair (complex.rs:108-113), and in an input whose residues are runs of atoms (every protein)
// a third of an atom's geometric neighbours are such atoms -- which otherwise ride through the gathers and the exact phase only to be dropped
// there (profiles/r04_emit_experiments.txt, section 11).  The staged chunk carries the residue word of every record (Sorted::rkey -> WaveLdsR::nkey);
// kbase = LDS byte address of the word of the run's test 0 for this lane, kh1 = the home atom's word - 1: the survivor is dropped when
// (its word - kh1) is 0, 1 or 2, i.e. when the words differ by at most 1 -- which, by the word's construction, happens only for atoms the
// reference's rule drops (a chain's ordinals stop 3 short of the next chain's; kh1 = kRkOff switches the rule off).  The exact phase still
// applies the rule itself on the real keys: this is an early exit, not the decision.
// The word of the NEXT round's survivor is requested before this round's append (one LDS round trip per round would otherwise sit on the chain):
// 11 vector + 6 scalar + 2 LDS instructions per round against 7 + 5 + 1.
DEVFN void compact_rounds_res_e(uint32_t &mask, uint32_t tag, uint32_t &qbyte, uint32_t qfull, uint32_t kbase, uint32_t kh1) {
    unsigned long long save, act;
    uint32_t t, lz, ent, bm, c, ka, kn, d;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_cmpx_ne_u32_e32 vcc, 0, %[mask]\n\t"
        "s_cbranch_vccz 2f\n\t"
        "v_ffbh_u32 %[lz], %[mask]\n\t"
        "v_lshl_add_u32 %[ka], %[lz], 2, %[kbase]\n\t"
        "ds_read_b32 %[kn], %[ka]\n\t"
        "1:\n\t"
        "v_add_u32 %[ent], %[tag], %[lz]\n\t"
        "v_lshrrev_b32 %[bm], %[lz], %[top]\n\t"
        "v_xor_b32 %[mask], %[mask], %[bm]\n\t"
        "v_ffbh_u32 %[lz], %[mask]\n\t"
        "v_lshl_add_u32 %[ka], %[lz], 2, %[kbase]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_u32 %[d], %[kn], %[kh1]\n\t"
        "v_cmp_lt_u32_e32 vcc, 2, %[d]\n\t"
        // (gfx9 hazard, "mixed use of VCC": a VALU instruction that reads vcc as a CONSTANT needs one wait state after the VALU compare that
        // wrote it, and nothing inserts it inside inline assembly -- v_mbcnt_lo right behind the compare ranked the lanes on the previous
        // round's mask.  The plain rounds get their wait state from the branch between v_cmpx and v_mbcnt_lo.)
        "ds_read_b32 %[kn], %[ka]\n\t"
        "s_bcnt1_i32_b64 %[c], vcc\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[qb]\n\t"
        "s_and_saveexec_b64 %[act], vcc\n\t"
        "ds_write_b32 %[t], %[ent]\n\t"
        "s_mov_b64 exec, %[act]\n\t"
        "s_lshl2_add_u32 %[qb], %[c], %[qb]\n\t"
        "v_cmpx_ne_u32_e32 vcc, 0, %[mask]\n\t"
        "s_cbranch_vccz 2f\n\t"
        "s_cmp_lt_u32 %[qb], %[qf]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "2:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 exec, %[save]"
        : [mask] "+v"(mask), [qb] "+s"(qbyte), [t] "=&v"(t), [lz] "=&v"(lz), [ent] "=&v"(ent), [bm] "=&v"(bm), [c] "=&s"(c), [save] "=&s"(save), [act] "=&s"(act),
          [ka] "=&v"(ka), [kn] "=&v"(kn), [d] "=&v"(d)
        : [tag] "v"(tag), [top] "s"(0x80000000u), [qf] "s"(qfull), [kbase] "v"(kbase), [kh1] "v"(kh1)
        : "vcc", "scc", "memory");
}

// mask = 2 * mask + (d2 < thr), as the sign bit of d2 - thr shifted in: a full-rate subtract and one v_alignbit_b32 instead of the half-rate
// v_cmp_le_f32 + v_addc_co_u32 pair of push_pass (pairs.inl) -- 8.2 -> ~4.6 cycles of the SIMD's vector issue per test
// (tests/microbench/issue_mix.hip: "cmp+addc" against "sub+alignbit").  A NaN may pass: the exact phase drops it.
DEVFN void push_sign_e(uint32_t &mask, float d2, float thr) {
    float t;
    asm("v_sub_f32 %1, %2, %3\n\tv_alignbit_b32 %0, %0, %1, 31" : "+v"(mask), "=&v"(t) : "v"(d2), "v"(thr));
}

struct ConstsE { double r2, s_hphob, s_ion, s_polar, s_cov_max; };  // wave-uniform: scalar registers

DEVFN double words_f64(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }

// Lane predicates as 64-bit scalar masks.  Left to the compiler, a predicate that is combined, balloted and branched on goes through
// v_cndmask 0/1 + v_cmp_ne round trips (two vector instructions and a hazard nop per use); as a scalar value it is combined on the
// scalar unit and only meets the vector pipe again as the condition of a select or as the exec mask of a store.
// Round 4: the compares write their scalar register pair directly (VOP3 form).  tests/microbench/issue_mix.hip: a scalar instruction
// holds the SIMD's scalar slot for 4 cycles -- as long as a half-rate vector instruction -- so the vcc + s_mov_b64 form of round 3 paid
// one such slot per mask for nothing.
typedef unsigned long long lmask;
#define ARP_LMASK_CMP(NAME, OP, TA, CA, TB, CB)                                                                  \
    DEVFN lmask NAME(TA a, TB b) {                                                                               \
        lmask m;                                                                                                 \
        asm(OP "_e64 %0, %1, %2" : "=s"(m) : CA(a), CB(b));                                                      \
        return m;                                                                                                \
    }
ARP_LMASK_CMP(lm_ge_f64_sv, "v_cmp_ge_f64", double, "s", double, "v")      // a (scalar) >= b
ARP_LMASK_CMP(lm_gt_f64_sv, "v_cmp_gt_f64", double, "s", double, "v")      // a (scalar) >  b
ARP_LMASK_CMP(lm_lt_u32_sv, "v_cmp_lt_u32", uint32_t, "s", uint32_t, "v")  // a (scalar) <  b
ARP_LMASK_CMP(lm_gt_u32_sv, "v_cmp_gt_u32", uint32_t, "s", uint32_t, "v")  // a (scalar) >  b
ARP_LMASK_CMP(lm_ne_u32, "v_cmp_ne_u32", uint32_t, "v", uint32_t, "v")
ARP_LMASK_CMP(lm_lt_u32_vv, "v_cmp_lt_u32", uint32_t, "v", uint32_t, "v")  // a < b
ARP_LMASK_CMP(lm_lt_u64, "v_cmp_lt_u64", unsigned long long, "v", unsigned long long, "v")
#undef ARP_LMASK_CMP
DEVFN uint32_t lm_select(lmask m, uint32_t if_set, uint32_t if_clear) {
    uint32_t d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if_clear), "v"(if_set), "s"(m));
    return d;
}
DEVFN bool lm_lane(lmask m, uint32_t lane) { return (m >> lane) & 1ull; }  // (rare paths only: a 64-bit vector shift)
DEVFN uint32_t lm_rank(lmask m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
DEVFN uint32_t lm_count(lmask m) {  // (as asm: the builtin popcount of an asm-produced mask comes back through a vector register)
    uint32_t c;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(c) : "s"(m) : "scc");
    return c;
}
// one 16-byte non-temporal record store per lane of m: scalar base + 32-bit lane offset (store_record, pairs.inl)
DEVFN void lm_store_records(lmask m, uint4 *base, uint32_t byte_off, const u32x4 &rec) {
    lmask save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx4 %2, %3, %4 nt\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(m), "v"(byte_off), "v"(rec), "s"(base) : "memory");
}

// The block allocator's common case without the compiler's help (round 3's form cost ~25 scalar instructions per batch: 64-bit
// sign-extension of the two readfirstlanes, selects, a loop header).  alloc_issue_e: lane 0 adds n to the block's LDS word {chunk << 32 |
// records used} -- exec is narrowed to lane 0 around the one instruction; the answer stays in lane 0's registers until the batch's fast tail (exact_finish_e) or alloc_finish_rt reads it.
DEVFN u32x2 alloc_issue_e(unsigned long long &state, uint32_t n) {
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned long long *)&state;
    u32x2 old, add = {n, 0u};
    lmask save;
    asm volatile("s_and_saveexec_b64 %1, 1\n\tds_add_rtn_u64 %0, %2, %3\n\ts_mov_b64 exec, %1" : "=&v"(old), "=&s"(save) : "v"(addr), "v"(add) : "memory");
    return old;
}
DEVFN unsigned long long u64_of(const u32x2 &v) { return ((unsigned long long)v.y << 32) | v.x; }
// the general placement of a batch's records: the run may cross a chunk end or the end of the caller's buffer (scratch until k_fixup)
DEVFN void store_batch_general(const Slots &sl, lmask m_valid, uint32_t n_rec, uint32_t rank, const u32x4 &rec, const EmitTarget &tg, unsigned long long *result,
                               uint32_t lane) {
    if (sl.n0 == n_rec && sl.pos0 + n_rec <= tg.capacity) {
        lm_store_records(m_valid, reinterpret_cast<uint4 *>(tg.out) + sl.pos0, rank << 4, rec);
    } else if (lm_lane(m_valid, lane)) {
        uint4 *d = emit_slot(tg, rank < sl.n0 ? sl.pos0 + rank : sl.pos1 + (rank - sl.n0), result);
        if (d) store_record(d, make_uint4(rec.x, rec.y, rec.z, rec.w));
    }
}

// The hole-free sequence of small inputs (k_emit's DIRECT kernels): a wave collects its records in a private LDS buffer and takes the places
// of a whole buffer at once from the global RECORD counter -- result[2], one returning atomic per flush; what is left at the end of the launch
// goes out in one flush per block.  The list then has no holes and the sequence no fix-up launch (11 us of a 60 us call); the atomics all hit
// one address (~11 ns each when they queue up), which is why this stops at the 4-wave kernels: 20 000 atoms are ~1 500 flushes.
constexpr uint32_t kStageRecords = 512;
struct StageRef { uint4 *buf; uint32_t n; };  // (n: wave-uniform)
DEVFN void stage_copy_e(const uint4 *buf, uint32_t n, unsigned long long pos, const EmitTarget &tg, unsigned long long *result, uint32_t lane) {
    wave_lds_fence();  // lanes read records other lanes wrote
    for (uint32_t k = lane; k < n; k += 64u) {
        uint4 *d = emit_slot(tg, pos + k, result);
        if (d) store_record(d, buf[k]);
    }
    wave_lds_fence();  // ... before the buffer is written again
}
// The hole-free sequence of inputs between the 4-wave kernels' range and the chip-filling ones (k_emit<.., STAGE>, round 5): the 12-wave blocks have
// no LDS to stage in, so a wave's records go to a region of its OWN in the engine's scratch block (kWaveStageRecords records, written with plain
// stores so that they stay in the L2 / Infinity Cache) and are copied to their final places when the wave has no task left -- the places of a
// whole block at once, from the global record counter (one returning atomic per block), or of one wave when its region fills up.  No chunks, no
// holes, no fix-up launch (13 us of an 80 us call at 10^5 atoms).  Only the wave itself ever touches its region: nothing has to become visible
// to another wave, and the copy waits for the wave's own stores (s_waitcnt vmcnt(0)) and reads the region past the vector L1 (device-scope loads:
// a region that was flushed once and refilled may still have its old lines there).  A final place beyond the caller's capacity is dropped, not
// spilled into the scratch block (which holds other waves' regions); the count goes on and tells the host how large a buffer the list needs.
constexpr uint32_t kWaveStageRecords = 1024;
DEVFN void stage_copy_g(const uint4 *buf, uint32_t n, unsigned long long pos, const EmitTarget &tg, uint32_t lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // four records per lane and trip, every load of a trip in flight before the first is waited for (the first version read one record per trip
    // through two device-scope atomic loads: ~2 us per trip, 16 us to empty a region -- the kernel was slower than the fix-up launch it saved)
    constexpr uint32_t kU = 4;
    for (uint32_t k0 = 0; k0 < n; k0 += 64u * kU) {
        u32x4 r[kU];
#pragma unroll
        for (uint32_t u = 0; u < kU; u++) {
            const uint4 *src = buf + min(k0 + 64u * u + lane, n - 1u);
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[u]) : "v"(src) : "memory");  // (sc1: past the vector L1, from the L2)
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
#pragma unroll
        for (uint32_t u = 0; u < kU; u++) {
            const uint32_t k = k0 + 64u * u + lane;
            if (k < n && pos + k < tg.capacity) store_record(reinterpret_cast<uint4 *>(tg.out) + (pos + k), make_uint4(r[u].x, r[u].y, r[u].z, r[u].w));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... before the region is written again
}
DEVFN void stage_flush_g(StageRef &sg, const EmitTarget &tg, unsigned long long *result, uint32_t lane) {
    const uint32_t n = __builtin_amdgcn_readfirstlane(sg.n);
    if (n == 0u) return;
    const Slots sl = alloc_direct(&result[2], n, lane);
    stage_copy_g(sg.buf, n, sl.pos0, tg, lane);
    sg.n = 0u;
}
// one plain (cached) 16-byte record store per lane of m: scalar base + 32-bit lane offset -- the staged records are read again (lm_store_records: nt)
DEVFN void lm_store_records_cached(lmask m, uint4 *base_v, uint32_t byte_off, const u32x4 &rec) {
    lmask save;
    // (the wave's region: wave-uniform by construction, but derived from threadIdx -- say so, the store wants its base in a scalar register pair)
    const uintptr_t bv = (uintptr_t)base_v;
    // (the builtin returns int: without the casts the low half is SIGN-extended into the high one -- a region above a 2 GB boundary then faults)
    uint4 *base = (uint4 *)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(bv >> 32)) << 32) | (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)bv));
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx4 %2, %3, %4\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(m), "v"(byte_off), "v"(rec), "s"(base) : "memory");
}
DEVFN void stage_flush_e(StageRef &sg, const EmitTarget &tg, unsigned long long *result, uint32_t lane) {
    const uint32_t n = __builtin_amdgcn_readfirstlane(sg.n);
    if (n == 0u) return;
    const Slots sl = alloc_direct(&result[2], n, lane);
    stage_copy_e(sg.buf, n, sl.pos0, tg, result, lane);
    sg.n = 0u;
}

// Phase 2 on the 64 queue entries that start at byte `qoff` of the wave's queue (FULL) or on its first `count` < 64 entries (!FULL, qoff = 0):
// home operands out of LDS, neighbour operands gathered (40 of the 48 bytes of the exact record).  Every lane computes everything -- the
// lanes beyond count on entry 0 (home lane 0, slot 0: in bounds) -- and only the stores are predicated.
// ONLY = ARP_FLAG_CONTACTS_ONLY: candidates without a row are dropped, so a record's place is only known once its rows are; the pairs
// a probe has to decide go to k_pairs_deferred, which emits them itself.  All candidates (!ONLY): every candidate is a record, so the
// output positions are requested from the block's allocator as soon as the cutoff test is in -- the answer travels while the rows are
// computed -- and a pair that needs a probe is written with kind 0 at its final position and listed {slots, position} for
// k_patch_deferred (pairs.inl).
// Round 4 (scalar diet): one rare-path branch for "a probe decides" and "the square root needs the exact routine" together, the
// allocator's common case in ~10 scalar instructions (the asm block at the end of exact_finish_e), masks straight out of the compares.
// The batch in two halves: exact_issue_e reads the batch's queue entries and sends the neighbour gathers on their way, exact_finish_e does
// the rest (counters after the scalar diet: 54 % of wave-cycles at s_waitcnt).
struct ExactRegs { uint32_t e; u32x4 bxy, bzp; unsigned long long kb; };
template <bool FULL>
DEVFN ExactRegs exact_issue_e(const WaveLdsE &w, const Sorted &so, uint32_t qoff, uint32_t count, uint32_t lane) {
    qoff = __builtin_amdgcn_readfirstlane(qoff); count = __builtin_amdgcn_readfirstlane(count);  // (wave-uniform by construction: say so)
    wave_lds_fence();  // lanes read entries other lanes wrote
    ExactRegs g;
    g.e = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(w.queue) + qoff + 4u * lane);
    if (!FULL) g.e = lane < count ? g.e : 0u;
    wave_lds_fence();
    uint32_t goff;  // 48 * (the entry's low 24 bits = the neighbour slot) in one instruction: the 24-bit multiply ignores the home-lane byte
    asm("v_mul_u32_u24 %0, %1, 48" : "=v"(goff) : "v"(g.e));
    const char *gp = reinterpret_cast<const char *>(so.fat) + (size_t)goff;
    g.bxy = *reinterpret_cast<const u32x4 *>(gp); g.bzp = *reinterpret_cast<const u32x4 *>(gp + 16);
    g.kb = *reinterpret_cast<const unsigned long long *>(gp + 32);
    return g;
}
// CHUNK == 1: the hole-free sequence of small inputs -- no block allocator, the batch's records go to the wave's staging buffer (sg) on the
// general path, and a pair a probe has to decide is decided HERE, by the lane that holds it (classify<true>, as k_patch_deferred does per
// list entry): these kernels run at two waves per SIMD anyway, the probes' registers cost them nothing, and a call without a probe pass is
// one launch shorter.  pe: the bounds the probes read (DIRECT only) + the clash / covalent bounds of the full level count (global memory).
struct ProbeParamsE { const double *s_clash, *s_cov, *s_vdw, *s_hacc; double s_ion, s_polar, s_hphob; };
template <bool FULL, bool ONLY, uint32_t CHUNK>
DEVFN void exact_finish_e(const ExactRegs &g, const ConstsE &K, const TablesE &tb, WaveLdsE &w, BlockLds &bl, uint32_t count, uint32_t slot0,
                          const EmitTarget &tg, uint32_t cap_chunks, unsigned long long *result, uint32_t lane, uint32_t wflags, uint32_t probe_bits, StageRef &sg,
                          const DevAtoms &in, const Sorted &so, const ProbeParamsE &pe) {
    constexpr bool DIRECT = CHUNK == 1u, STAGE = CHUNK == 2u;  // (CHUNK 1 / 2: the hole-free sequences -- records staged per wave in LDS / in the wave's scratch region)
    count = __builtin_amdgcn_readfirstlane(count);
    const uint32_t e = g.e, hl = e >> kESlotBits, nb = e & kESlotMask;
    const u32x4 bxy = g.bxy, bzp = g.bzp;
    const unsigned long long kb = g.kb;
    // (two addresses, hl * 16 and hl * 8 from the wave's base, the arrays as immediate offsets)
    const char *wb = reinterpret_cast<const char *>(&w);
    const u32x4 axy = *reinterpret_cast<const u32x4 *>(wb + offsetof(WaveLdsE, hxy) + 16u * hl), azp = *reinterpret_cast<const u32x4 *>(wb + offsetof(WaveLdsE, hzp) + 16u * hl);
    const unsigned long long ka = *reinterpret_cast<const unsigned long long *>(wb + offsetof(WaveLdsE, hkey) + 8u * hl);
    // pdbtbx Atom::distance before the sqrt, f64, the reference's operation order, no contraction
    const double s = sq_dist(words_f64(axy.x, axy.y), words_f64(axy.z, axy.w), words_f64(azp.x, azp.y), words_f64(bxy.x, bxy.y), words_f64(bxy.z, bxy.w),
                             words_f64(bzp.x, bzp.y));
    const uint32_t pa = azp.z, pb = bzp.z;
    // should_compare_entities for both orientations (complex.rs:76-131); at most one can hold
    lmask m_ok, m_swap;
    if (wflags & kWaveAllBoth) {  // (wave-uniform: a scalar branch)  every atom in both chain sets: the ligand is the atom with the smaller key
        const uint32_t d1 = (uint32_t)kb - (uint32_t)ka + 1u;                   // residue ordinals of one chain: |difference| >= 2 (:113)
        m_ok = lm_lt_u32_sv(2u, d1) | lm_ne_u32((uint32_t)(ka >> 32), (uint32_t)(kb >> 32));  // another chain: always (:124-129); models never meet in the grid
        m_swap = lm_lt_u64(kb, ka);
    } else {  // chain groups: orient() of kernels.hip on lane masks (every comparison lands in a scalar register pair)
        const uint32_t ca = (uint32_t)(ka >> 32), cb = (uint32_t)(kb >> 32);
        const lmask same_model = ~0ull;  // :96-98 is the grid's: a slot window never holds an atom of another model (arp_internal.h Fat::crm)
        const lmask same_chain = ~lm_ne_u32(ca, cb);
        const lmask ab_chain = lm_lt_u32_vv((uint32_t)ka + 1u, (uint32_t)kb), ba_chain = lm_lt_u32_vv((uint32_t)kb + 1u, (uint32_t)ka);  // :108,:113
        const uint32_t lr = pa & pb;                                                                            // bit 24: both ligand, bit 25: both receptor
        const lmask both = lm_lt_u32_sv(0u, lr & (lr >> 1) & kPwLigand);                                        // :124-129
        const lmask ab_cross = ~(both & lm_lt_u32_vv(cb, ca)), ba_cross = ~(both & lm_lt_u32_vv(ca, cb));
        const lmask aL = lm_lt_u32_sv(0u, pa & kPwLigand), aR = lm_lt_u32_sv(0u, pa & kPwReceptor);
        const lmask bL = lm_lt_u32_sv(0u, pb & kPwLigand), bR = lm_lt_u32_sv(0u, pb & kPwReceptor);
        const lmask o1 = same_model & aL & bR & ((same_chain & ab_chain) | (~same_chain & ab_cross));
        const lmask o2 = same_model & bL & aR & ((same_chain & ba_chain) | (~same_chain & ba_cross));
        m_ok = o1 | o2; m_swap = o2 & ~o1;
    }
    lmask m_valid = lm_ge_f64_sv(K.r2, s) & m_ok;  // rstar: inclusive
    if (!FULL) m_valid &= (1ull << (count & 63u)) - 1ull;  // (!FULL: count < 64)
    uint32_t n_rec = ONLY ? 0u : lm_count(m_valid);
    u32x2 a_old = {0u, 0u};
    if (!ONLY) {
        if (__builtin_expect(n_rec == 0u, 0)) return;
        if (!DIRECT && !STAGE) a_old = alloc_issue_e(bl.alloc_state, n_rec);
    }
    // distance levels: Le against the element pair's bounds {vdw, cov, clash}, Lg against the fixed ones {4.5, 4.0, 3.5}; L = 4 Le + Lg.
    // Only the van-der-Waals bound on the common path -- a distance below the covalent or the clash bound of ITS
    // element pair is first of all below the largest covalent bound of any pair (K.s_cov_max, a scalar), which hardly any candidate is
    // (the atoms of one residue and of sequence neighbours never pair, complex.rs:108-113); those batches recompute the levels in full on
    // the general path below.  Two LDS reads and four half-rate vector instructions less per batch.
    const uint32_t eix = (pa & 0xF0u) | (pb & 0x0Fu);
    const double t_vdw = tb.s_vdw[eix];
    uint32_t L = 0;
    lmask m_close = 0ull;
    auto levels_full = [&]() {
        const double t_clash = pe.s_clash[eix], t_cov = pe.s_cov[eix];  // (global memory, 2 x 2 KB that sit in the vector L1: rare path)
        uint32_t Lf = 0;
        asm("v_cmp_lt_f64_e32 vcc, %[s], %[tv]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc\n\t"
            "v_cmp_lt_f64_e32 vcc, %[s], %[tc]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc\n\t"
            "v_cmp_lt_f64_e32 vcc, %[s], %[tx]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc\n\t"
            "v_lshlrev_b32_e32 %[L], 2, %[L]\n\t"
            "v_cmp_gt_f64_e32 vcc, %[k45], %[s]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc\n\t"
            "v_cmp_gt_f64_e32 vcc, %[k40], %[s]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc\n\t"
            "v_cmp_gt_f64_e32 vcc, %[k35], %[s]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc"
            : [L] "+v"(Lf)
            : [s] "v"(s), [tv] "v"(t_vdw), [tc] "v"(t_cov), [tx] "v"(t_clash), [k45] "s"(K.s_hphob), [k40] "s"(K.s_ion), [k35] "s"(K.s_polar)
            : "vcc");
        return Lf;
    };
    {
        asm("v_cmp_lt_f64_e32 vcc, %[s], %[tv]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc\n\t"
            "v_lshlrev_b32_e32 %[L], 2, %[L]\n\t"
            "v_cmp_gt_f64_e32 vcc, %[k45], %[s]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc\n\t"
            "v_cmp_gt_f64_e32 vcc, %[k40], %[s]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc\n\t"
            "v_cmp_gt_f64_e32 vcc, %[k35], %[s]\n\t"
            "v_addc_co_u32_e32 %[L], vcc, 0, %[L], vcc"
            : [L] "+v"(L)
            : [s] "v"(s), [tv] "v"(t_vdw), [k45] "s"(K.s_hphob), [k40] "s"(K.s_ion), [k35] "s"(K.s_polar)
            : "vcc");
        m_close = lm_gt_f64_sv(K.s_cov_max, s) & m_valid;
    }
    // contacts only: the record count depends on the kinds, so a batch with a candidate inside some covalent band counts its levels in full right
    // here (a wave-uniform branch, hardly ever taken)
    if (ONLY && m_close) { L = levels_full(); m_close = 0ull; }
    // W = (Pa & Qb) | (Pb & Qa): P is byte 1 of the pair word, Q byte 2
    uint32_t w1, w2;
    asm("v_and_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n\t"
        "v_and_b32_sdwa %1, %3, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2"
        : "=&v"(w1), "=&v"(w2) : "v"(pa), "v"(pb));
    uint32_t t = tb.lut[(L << 7) | w1 | w2];
    // a probe decides: bit 30 & (either residue carries hydrogens), bit 29 & (residue tables present); bit 31 is never set
    lmask m_rare = lm_lt_u32_sv(0x1FFFFFFFu, t & (pa | pb | probe_bits));  // (& m_valid below)
    uint32_t kind = t & 0x1FFFFFFFu;
    if (ONLY) m_valid &= lm_lt_u32_sv(0u, kind) | m_rare;  // no-interaction candidates are dropped
    lmask m_defer = m_rare & m_valid;
    // (f32) of the correctly rounded f64 sqrt (kernels.hip dist_f32), the rare exact path behind a wave-uniform branch
    const double r = (double)__frsqrt_rn((float)s);
    const double y0 = s * r, hr = 0.5 * r;
    double y = __fma_rn(__fma_rn(-y0, y0, s), hr, y0);
    const uint32_t low = (uint32_t)__double_as_longlong(y) & 0x1FFFFFFFu;  // the 29 bits a cast to f32 drops
    // s outside [2^-100, 2^100) (zero, non-finite), or y within 2048 f64 ulps of an f32 rounding boundary
    const lmask m_exact = m_valid & (lm_lt_u32_sv(0x46300000u - 0x39B00000u - 1u, (uint32_t)__double2hiint(s) - 0x39B00000u) |
                                     lm_gt_u32_sv(4097u, low - (0x10000000u - 2048u)));
    if (ONLY) {  // the records that stay are known: ask for their places now, the answer is read after the rare paths
        n_rec = lm_count(m_valid & ~m_defer);
        if (n_rec && !DIRECT && !STAGE) a_old = alloc_issue_e(bl.alloc_state, n_rec);
    }
    u32x4 rec;
    rec.x = lm_select(m_swap, bzp.w, azp.w); rec.y = lm_select(m_swap, azp.w, bzp.w);
    uint32_t general;  // (wave-uniform, and opaque to the optimiser in both arms: as a bool it comes back as a lane mask + three scalar instructions per use)
    if (!DIRECT && __builtin_expect((m_exact | m_defer | m_close) == 0ull, 1)) {
        rec.z = __float_as_uint((float)y);
        rec.w = kind;
        if (ONLY) { if (n_rec == 0u) return; }
        if constexpr (STAGE) {  // the wave's own region: the next n_rec places, no allocator at all
            if (__builtin_expect(sg.n + n_rec > kWaveStageRecords, 0)) stage_flush_g(sg, tg, result, lane);
            lm_store_records_cached(m_valid, sg.buf, (sg.n + lm_rank(m_valid)) << 4, rec);
            sg.n += n_rec;
            return;
        }
        // The allocator's common case, hand-scheduled (left to the compiler: ~25 scalar instructions of selects and flag words): the n records
        // fit the block's current chunk (used + n <= CHUNK) and the chunk lies wholly inside the first 2^32 bytes of the caller's buffer (chunk <
        // cap_chunks) -> one run, scalar base + the 32-bit byte offset (position + rank) * 16, stored under the valid mask as exec.
        const uint32_t rank = lm_rank(m_valid);
        uint32_t t0, t1, off;
        lmask save;
        asm volatile(
            "v_readfirstlane_b32 %[t0], %[olo]\n\t"
            "v_readfirstlane_b32 %[t1], %[ohi]\n\t"
            "s_add_u32 %[g], %[t0], %[n]\n\t"
            "s_cmp_le_u32 %[g], %[chunk]\n\t"
            "s_cselect_b32 %[g], %[t1], -1\n\t"
            "s_cmp_lt_u32 %[g], %[cap]\n\t"
            "s_cselect_b32 %[g], 0, 1\n\t"
            "s_cbranch_scc0 9f\n\t"
            "s_lshl_b32 %[t1], %[t1], %[shift]\n\t"
            "s_add_u32 %[t1], %[t1], %[t0]\n\t"
            "v_add_lshl_u32 %[off], %[rank], %[t1], 4\n\t"
            "s_and_saveexec_b64 %[save], %[mv]\n\t"
            "global_store_dwordx4 %[off], %[rec], %[base] nt\n\t"
            "s_mov_b64 exec, %[save]\n\t"
            "9:"
            : [g] "=&s"(general), [t0] "=&s"(t0), [t1] "=&s"(t1), [off] "=&v"(off), [save] "=&s"(save)
            : [olo] "v"(a_old.x), [ohi] "v"(a_old.y), [n] "s"(n_rec), [chunk] "s"(CHUNK), [cap] "s"(cap_chunks), [shift] "n"(chunk_shift_of(CHUNK)), [rank] "v"(rank),
              [mv] "s"(m_valid), [rec] "v"(rec), [base] "s"(tg.out)
            : "scc", "memory");
    } else {
        asm volatile("" ::: "memory");  // keep this a branch (no speculation of the long sequence)
        if (m_exact) { if (lm_lane(m_exact, lane)) y = sqrt(s); }
        rec.z = __float_as_uint((float)y);
        if (!ONLY && m_close) {  // a candidate inside the covalent range of some element pair: the levels in full, and what follows from them
            t = tb.lut[(levels_full() << 7) | w1 | w2];
            kind = t & 0x1FFFFFFFu;
            m_defer = lm_lt_u32_sv(0x1FFFFFFFu, t & (pa | pb | probe_bits)) & m_valid;
        }
        asm volatile("s_mov_b32 %0, 1" : "=s"(general));
    }
    if (__builtin_expect(general != 0u, 0)) {  // ---- chunk crossing / refill / scratch / a probe decides: the general placement ----
        if (DIRECT && m_defer) {  // the probes, inline (rare: a donor..acceptor pair of a residue with hydrogens, a CYS SG pair in the covalent band)
            if (lm_lane(m_defer, lane)) {
                const Fat a = so.fat[slot0 + hl], b = so.fat[nb];
                const double s2 = sq_dist(a.x, a.y, a.z, b.x, b.y, b.z);
                kind = classify<true, ProbeParamsE>(in, pe, s2, a, b, orient(a, b) == 2, result);
            }
            if (ONLY) { m_valid &= ~m_defer | lm_lt_u32_sv(0u, kind); n_rec = lm_count(m_valid); }  // a probe-decided pair without a row is dropped like any other
            m_defer = 0ull;
        }
        if (ONLY) {
            if (m_defer) {  // candidates whose rules need a probe go to the deferred pass (k_pairs_deferred), as global slot pairs
                const Slots ds = alloc_chunked<kDeferChunk>(bl.defer_state, &result[3], (uint32_t)__popcll(m_defer), lane);
                if (lm_lane(m_defer, lane)) {
                    const uint32_t dr = lm_rank(m_defer);
                    const unsigned long long p = dr < ds.n0 ? ds.pos0 + dr : ds.pos1 + (dr - ds.n0);
                    if (p < tg.defer_cap) tg.defer_list[p] = make_uint2(slot0 + hl, nb); else atomicOr(&result[1], 8ull);
                }
                m_valid &= ~m_defer;
            }
            if (n_rec == 0u) return;
        }
        if (STAGE && m_defer) {  // launched on the engine's memo that this input defers nothing, and it does after all: the host repeats the call with the
                                 // chunked sequence and its probe pass (status bit 128, as k_fixup raises it for a skipped pass); these records do not matter
            if (lane == 0u) atomicOr(&result[1], 128ull);
            if (ONLY) n_rec = lm_count(m_valid);
            m_defer = 0ull;
        }
        if (DIRECT) {
            if (sg.n + n_rec > kStageRecords) stage_flush_e(sg, tg, result, lane);
            rec.w = kind;
            if (lm_lane(m_valid, lane)) sg.buf[sg.n + lm_rank(m_valid)] = make_uint4(rec.x, rec.y, rec.z, rec.w);
            sg.n += n_rec;
            return;
        }
        if (STAGE) {
            if (n_rec == 0u) return;
            if (sg.n + n_rec > kWaveStageRecords) stage_flush_g(sg, tg, result, lane);
            rec.w = kind;
            if (lm_lane(m_valid, lane)) sg.buf[sg.n + lm_rank(m_valid)] = make_uint4(rec.x, rec.y, rec.z, rec.w);
            sg.n += n_rec;
            return;
        }
        const Slots sl = alloc_finish<CHUNK>(bl.alloc_state, &result[2], n_rec, lane, u64_of(a_old));
        const uint32_t rank = lm_rank(m_valid);
        if (!ONLY && m_defer) {  // the probe pass patches these kinds in place: it is told the records' final positions
            const Slots ds = alloc_chunked<kDeferChunk>(bl.defer_state, &result[3], 2u * (uint32_t)__popcll(m_defer), lane);  // (even counts: a pair never straddles a chunk)
            if (lm_lane(m_defer, lane)) {
                const unsigned long long pos = rank < sl.n0 ? sl.pos0 + rank : sl.pos1 + (rank - sl.n0);
                const uint32_t dr = 2u * lm_rank(m_defer);
                const unsigned long long p = dr < ds.n0 ? ds.pos0 + dr : ds.pos1 + (dr - ds.n0);
                if (p + 1ull < tg.defer_cap) { tg.defer_list[p] = make_uint2(slot0 + hl, nb); tg.defer_list[p + 1ull] = make_uint2((uint32_t)pos, (uint32_t)(pos >> 32)); }
                else atomicOr(&result[1], 8ull);
                kind = 0u;
            }
        }
        rec.w = kind;
        store_batch_general(sl, m_valid, n_rec, rank, rec, tg, result, lane);
    }
}

// WAVES per block: kEWaves; 4 for the smallest inputs (below 320 tasks), whose few tasks then spread over more CUs
// SPLIT: 1, or 4 = a task's five window kinds are shared out over four waves ({0, 1}, {2}, {3}, {4}), or 8 = two waves per kind set on
// alternate 32-test runs: a small input has few tasks and each is a long chain of dependent round trips, so more waves on a
// fraction of the chain each is what shortens the launch
// DIRECT (with SPLIT 8, the smallest inputs): no allocation chunks -- a batch takes its places from the global record counter, the list has
// no holes and the launch sequence no fix-up kernel (stage_flush_e above; engine.cpp finish_result)
struct StageLdsE { uint4 rec[4][kStageRecords]; uint32_t wave_n[4]; unsigned long long base; };
struct NoStageLdsE { uint4 rec[1][1]; uint32_t wave_n[1]; unsigned long long base; };
struct ScratchStageLdsE { uint4 rec[1][1]; uint32_t wave_n[kEWaves]; unsigned long long base; };  // STAGE: the records are in the scratch block, only the block-end sums here
// RES: the residue rule is applied to every prefilter survivor before it is queued (compact_rounds_res_e) -- the launcher's choice for inputs whose
// residues are runs of atoms; the result is the same list either way
// STAGE: the hole-free sequence of the 12-wave kernels with the four-way task split (stage_copy_g above): launched only when the engine's memo says the input
// defers nothing to the probe pass
template <int WAVES, int SPLIT, bool ONLY, bool DIRECT = false, bool RES = false, bool STAGE = false>
__global__ __launch_bounds__(WAVES * 64, WAVES == kEWaves ? kEWavesPerSimd : 2) void k_emit(DevAtoms in, const GridParams *gp, const DevParams *dprm, const uint32_t *cell_start, Sorted so,
                                                                                           EmitTarget tg, ulonglong2 *hole_list, uint32_t *task_ctr, unsigned long long *result) {
    // Inputs that do not fill the chip emit a few thousand records per block, or a few hundred: a 4096-record chunk per block would leave
    // holes as large as the list itself for the fix-up to close.  (Not smaller than this: every chunk costs a returning atomic on the one
    // global counter, ~11 ns each when they queue up -- 256-record chunks made the kernel 2.4x slower on 10^5 atoms and 30 % slower on 2x10^4.)
    // The 4-wave kernels' inputs (DIRECT: below 320 tasks) go without chunks altogether: the records are staged per wave and flushed to places
    // taken from the global record counter (stage_flush_e): no holes, no fix-up launch.
    static_assert(!DIRECT || WAVES == 4, "the hole-free sequence is the 4-wave kernels'");
    static_assert(!STAGE || (!DIRECT && SPLIT == 4), "the scratch-staged sequence is the 12-wave kernels' with the task split");
    constexpr uint32_t kChunkE = DIRECT ? 1u : (STAGE ? 2u : (SPLIT == 4 ? kSmallChunkRecords : kChunkRecords));
    static_assert(DIRECT || WAVES == kEWaves, "the 4-wave kernels are DIRECT");
    static_assert(!(RES && DIRECT), "the residue-rule rounds are the 12-wave kernels'");
    __shared__ TablesE tb;
    __shared__ typename std::conditional<RES, WaveLdsR, WaveLdsN>::type wl[WAVES];
    __shared__ BlockLds bl;
    __shared__ typename std::conditional<DIRECT, StageLdsE, typename std::conditional<STAGE, ScratchStageLdsE, NoStageLdsE>::type>::type stg;
    // (the grid and parameter words first: their loads travel together with the table's instead of behind the barrier)
    const uint32_t nx = gp->nx, ny = gp->ny, nzt = gp->nzt, kx = gp->kx, n_heavy = gp->n_heavy, n_tasks = gp->n_tasks * (uint32_t)SPLIT;  // (wave-tasks)
    const uint32_t sy = gp->sy_shift;  // (wave-uniform) the cell rows' order: arp_internal.h grid_row
    const uint32_t wflags = gp->all_both ? kWaveAllBoth : 0u;
    const uint32_t rk_off = RES ? gp->rk_bad : 0u;  // (wave-uniform) an ordinal or a chain rank did not fit the residue words: no early rejection
    const ConstsE K{dprm->r2, dprm->s_hphob, dprm->s_ion, dprm->s_polar, dprm->s_cov_max};
    const double r2m = gp->r2m;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // DIRECT: wave-task = the wave's index in the launch (one each: the launcher sends as many waves as there are wave-tasks), so the home
    // records can be asked for right here, with the tables -- one round trip less in a call that is a short chain of them.  (Slots past the
    // last heavy atom read whatever the workspace holds there: allocated, and masked by `have` below.)
    float4 home_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    Fat fat_pre{};
    if (DIRECT) {
        // (clamped to what the workspace allocates behind the last atom, engine.cpp ensure_workspace: n_cap + 64 records; the launch rounds the
        // wave-tasks up to whole blocks, so the last block's spare waves would otherwise read up to 62 records further)
        const uint32_t a0 = min(((blockIdx.x * (uint32_t)WAVES + wave) / (uint32_t)SPLIT) * 64u + lane, in.n + 63u);
        home_pre = so.rec[a0]; fat_pre = so.fat[a0];
    }
    load_tables_e(tb, dprm);
    if (threadIdx.x == 0) {
        bl.alloc_state = kAllocEmpty | kChunkE;  // "exhausted": the first allocation fetches a chunk
        bl.defer_state = kAllocEmpty | kDeferChunk;
        bl.chunk_shift = chunk_shift_of(kChunkE);
    }
    __syncthreads();
    const uint32_t probe_bits = in.n_res != 0u ? (1u << 29) : 0u;  // residue tables present: CYS SG pairs in the covalent band get their dihedral probe
    // chunks that lie wholly inside the caller's buffer: a batch placed in one of them needs no further capacity test (exact_finish_e's fast tail)
    // (and inside its first 2^32 bytes: the fast path addresses with a 32-bit byte offset; what lies beyond takes the general path)
    const uint32_t cap_chunks = (uint32_t)min(tg.capacity >> chunk_shift_of(kChunkE), (unsigned long long)((1u << 28) / kChunkE));
    WaveLdsE &w = wl[wave].e;
    uint32_t nkey_lds = 0;  // RES: LDS byte address of the staged chunk's residue words
    if constexpr (RES) nkey_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)wl[wave].nkey);
    StageRef sg{STAGE ? reinterpret_cast<uint4 *>(tg.scratch) + (size_t)(blockIdx.x * (uint32_t)WAVES + wave) * kWaveStageRecords : stg.rec[DIRECT ? wave : 0u], 0u};
    const ProbeParamsE pe{dprm->s_clash, dprm->s_cov, tb.s_vdw, dprm->s_hacc, K.s_ion, K.s_polar, K.s_hphob};
    // task distribution as in k_pairs: block group (b mod 8) = one XCD = one contiguous eighth of the tasks, static first task per wave
    const uint32_t n_groups = min(8u, gridDim.x), group = blockIdx.x % n_groups;
    const uint32_t g_lo = (uint32_t)(((unsigned long long)n_tasks * group) / n_groups), g_hi = (uint32_t)(((unsigned long long)n_tasks * (group + 1u)) / n_groups);
    uint32_t *ctr = task_ctr + (kEmit * 8 + group) * kTaskCtrStride;
    const uint32_t queue_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)w.queue);
    const uint32_t group_waves = ((gridDim.x - group + n_groups - 1u) / n_groups) * WAVES;
    uint32_t t = DIRECT ? blockIdx.x * (uint32_t)WAVES + wave : g_lo + (blockIdx.x / n_groups) * WAVES + wave;
    const uint32_t t_end = DIRECT ? n_tasks : g_hi;
#pragma unroll 1
    while (t < t_end) {
        constexpr uint32_t kSubs = SPLIT == 8 ? 2u : 1u;  // SPLIT == 8: two waves per kind set, on alternate 32-test runs
        const uint32_t slot0 = (t / (uint32_t)SPLIT) * 64u, part = (t % (uint32_t)SPLIT) / kSubs, sub = (t % (uint32_t)SPLIT) % kSubs;
        const int k_lo = SPLIT == 1 ? 0 : (part == 0u ? 0 : (int)part + 1), k_hi = SPLIT == 1 ? 5 : (int)part + 2;  // window kinds of this wave-task
        const uint32_t a = slot0 + lane;  // this lane's home slot
        const bool have = a < n_heavy;
        float4 home = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t cx = 0, cy = 0, cz = 0;
        uint32_t kh1 = kRkOff;  // RES: the home atom's residue word - 1
        wave_lds_fence();  // the previous task's batches are done with the home records
        {
            u32x4 hxy = {0u, 0u, 0u, 0u}, hzp = {0u, 0u, 0u, 0u};
            unsigned long long hkey = 0ull;
            if (have) {
                home = DIRECT ? home_pre : so.rec[a];
                const Fat &f = *(DIRECT ? &fat_pre : &fat_at<false>(so.fat, a));
                const double fx = f.x, fy = f.y, fz = f.z;
                hxy = u32x4{(uint32_t)__double2loint(fx), (uint32_t)__double2hiint(fx), (uint32_t)__double2loint(fy), (uint32_t)__double2hiint(fy)};
                hzp = u32x4{(uint32_t)__double2loint(fz), (uint32_t)__double2hiint(fz), f.pw, f.orig};
                hkey = ((unsigned long long)f.crm << 32) | f.res_ord;
                if (RES && !rk_off) kh1 = ((f.crm & kRkChainMax) << kRkOrdBits) + (f.res_ord & ((1u << kRkOrdBits) - 1u)) - 1u;
                const uint32_t c = f.cell;
                const uint32_t row = c / nx;
                cx = c - row * nx; grid_row_decode(row, ny, nzt, sy, cy, cz);
            }
            w.hxy[lane] = hxy; w.hzp[lane] = hzp; w.hkey[lane] = hkey;
        }
        const uint32_t xlo = cx > kx ? cx - kx : 0u, xhi = min(cx + kx, nx - 1);
        // all five slot windows of this lane up front: ten independent loads in flight instead of five round trips
        uint32_t wlo[5] = {0, 0, 0, 0, 0}, whi[5] = {0, 0, 0, 0, 0};
        if (have) {
            wlo[0] = a + 1; whi[0] = cell_start[grid_row(cy, cz, ny, nzt, sy) * nx + xhi + 1];
#pragma unroll
            for (int k = 1; k < 5; k++) {
                const int dy = (k == 1) ? 1 : (k - 3);
                const uint32_t zz = cz + (k == 1 ? 0u : 1u);
                const int yy = (int)cy + dy;
                if (yy >= 0 && yy < (int)ny && zz < nzt) {
                    const uint32_t r = grid_row((uint32_t)yy, zz, ny, nzt, sy) * nx;
                    wlo[k] = cell_start[r + xlo]; whi[k] = cell_start[r + xhi + 1];
                }
            }
        }
        // per-lane constants of the prefilter: -2 h (exact in f32) and the threshold r2m - |h|^2, rounded up (DESIGN.md "Prefilter margin")
        const float3 hm2 = make_float3(-2.0f * home.x, -2.0f * home.y, -2.0f * home.z);
        // (one ulp above the rounded-up threshold: the test below is the SIGN of acc - thr, which a tie would fail)
        const float thr_ru = __double2float_ru(r2m - ((double)home.x * home.x + (double)home.y * home.y + (double)home.z * home.z));
        const float thr = thr_ru > 0.0f ? __uint_as_float(__float_as_uint(thr_ru) + 1u) : (thr_ru < 0.0f ? __uint_as_float(__float_as_uint(thr_ru) - 1u) : 1e-37f);
        const uint32_t lane_tag = lane << kESlotBits;
        uint32_t qbyte = queue_lds;  // LDS byte address of the queue tail (wave-uniform): the phase-1 survivors waiting in w.queue; drained at the end of the task
#pragma unroll 1
        for (int k = k_lo; k < k_hi; k++) {
            uint32_t lo = wlo[0], hi = whi[0];
#pragma unroll
            for (int j = 1; j < 5; j++) if (k == j) { lo = wlo[j]; hi = whi[j]; }
            const bool nonempty = lo < hi;
            const uint32_t Lw = wave_min_u32(nonempty ? lo : 0xFFFFFFFFu), Hw = wave_max_u32(nonempty ? hi : 0u);
            if (Lw >= Hw) continue;
#pragma unroll 1
            for (uint32_t cs = Lw; cs < Hw; cs += kEChunk) {
                const uint32_t ce = min(cs + kEChunk, Hw);
                const uint32_t j0 = max(lo, cs), j1 = min(hi, ce);
                const uint32_t len = (nonempty && j1 > j0) ? j1 - j0 : 0u;
                if (!__any(len != 0u)) {  // no lane's window reaches into this chunk: on to the first slot any lane still needs (lanes of one task can sit in
                    // rows whose neighbour rows lie far apart in the cell order -- a whole y strip apart at a strip's edge, arp_internal.h grid_row)
                    const uint32_t need = wave_min_u32((nonempty && hi > ce) ? max(lo, ce) : 0xFFFFFFFFu);
                    if (need >= Hw) break;
                    cs = need - kEChunk;  // (>= cs: the chunk was a whole one, or no lane would be left; the loop's increment follows)
                    continue;
                }
                wave_lds_fence();  // previous chunk fully consumed
                // Staging: both halves of the chunk requested before either is waited for (one round trip per chunk, not two), from clamped
                // addresses (the pad of a short chunk holds copies of its last record: never inside a lane's window).
                // (A software prefetch of the chunk's EXACT records here -- one word of each, coalesced, so that the survivors' gathers find
                // their lines in the L2 -- was built and measured in round 4: 161-162 us with and without; the gathers' cost is their
                // address work, 21 us per million wave-instructions, not the misses.  profiles/r04_emit_experiments.txt.)
                {
                    static_assert(kEChunk == 128u, "two staging loads per lane");
                    const uint32_t p0 = min(cs + lane, ce - 1u), p1 = min(cs + lane + 64u, ce - 1u);
                    const float4 r0 = so.rec[p0], r1 = so.rec[p1];
                    if constexpr (RES) {
                        const uint32_t k0 = so.rkey[p0], k1 = so.rkey[p1];
                        wl[wave].nkey[lane] = k0; wl[wave].nkey[lane + 64u] = k1;
                    }
                    w.nrec[lane] = r0; w.nrec[lane + 64u] = r1;
                }
                wave_lds_fence();
                const uint32_t off = len ? j0 - cs : 0u;
                uint32_t it0 = sub * kEAcc;  // (wave-uniform)  SPLIT == 8: the partner wave takes every other run of kEAcc tests
                bool more = kSubs == 1u || __any(it0 < len);
#pragma unroll 1
                while (more) {
                    // Phase 1, one run: up to kEAcc prefilter tests per lane in groups of kEGroup, results pushed into the lane's mask word.
                    // Every lane reads from its own window start onwards, ALL addresses of the run being immediate offsets from one base: a
                    // lane whose window is exhausted reads on past it (other records, the wave's other arrays, at worst beyond the block's LDS
                    // allocation) and its bits are dropped below -- no per-group clamp, select or address arithmetic.  The hardware contract
                    // this leans on is that an LDS read past the allocation never faults; the VALUE is unspecified (stale words of the
                    // allocation granule's padding just past the end, 0 further out: tests/lds_oob, a -m gpu test) and never used.
                    const uint32_t acc0 = it0;
                    uint32_t mask = 0, nacc = 0;
                    const float4 *win = w.nrec + (off + it0);
#pragma unroll
                    for (uint32_t g = 0; g < kEAcc / kEGroup; ++g) {
#pragma unroll
                        for (uint32_t u0 = 0; u0 < kEGroup; u0 += kReadAhead) {
                            float rx[kReadAhead], ry[kReadAhead], rz[kReadAhead], rw[kReadAhead];
#pragma unroll
                            for (uint32_t u = 0; u < kReadAhead; ++u) { const float4 r = win[g * kEGroup + u0 + u]; rx[u] = r.x; ry[u] = r.y; rz[u] = r.z; rw[u] = r.w; }
                            // |n|^2 - 2 n.h against thr = r2m - |h|^2; the FMAs link-major over the tests in flight so that neighbours are independent
                            float acc[kReadAhead];
#pragma unroll
                            for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(rx[u], hm2.x, rw[u]);
#pragma unroll
                            for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(ry[u], hm2.y, acc[u]);
#pragma unroll
                            for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(rz[u], hm2.z, acc[u]);
#pragma unroll
                            for (uint32_t u = 0; u < kReadAhead; ++u) push_sign_e(mask, acc[u], thr);
                        }
                        nacc = (g + 1u) * kEGroup;
                        more = __any(it0 + nacc < len);
                        if (!more) break;
                    }
                    it0 += nacc;
                    // left-align (test q of the run -> bit 31 - q) and drop the tests past the lane's window end: keep the top min(rem, 32) bits
                    // = the low word of 0xFFFFFFFF00000000 >> rem
                    const uint32_t rem = min(len > acc0 ? len - acc0 : 0u, 32u);
                    mask = (mask << (32u - nacc)) & (uint32_t)(0xFFFFFFFF00000000ull >> rem);
                    const uint32_t tag = lane_tag + (cs + off + acc0);  // (bit 31 - lz <-> test q = lz <-> neighbour slot cs + off + acc0 + q)
                    const uint32_t kbase = nkey_lds + 4u * (off + acc0);  // RES: ... <-> residue word nkey[off + acc0 + q]
                    auto compact = [&](uint32_t &qb) {
                        if constexpr (RES) compact_rounds_res_e(mask, tag, qb, queue_lds + 256u, kbase, kh1);
                        else compact_rounds_e(mask, tag, qb, queue_lds + 256u);
                    };
                    {
                        uint32_t qb = __builtin_amdgcn_readfirstlane(qbyte);  // (wave-uniform by construction: say so)
                        compact(qb);
                        while (qb >= queue_lds + 256u) {  // a full batch: the 64 entries at the tail; then the rest of the run's survivors
                            qb -= 256u;
                            // (Finishing the batch only after the next compaction rounds -- the gathers in flight meanwhile -- was measured: no
                            // gain, 162 us either way; the compiler then parks the next prefilter run on vmcnt(0) for a register it sees reused.)
                            const ExactRegs g = exact_issue_e<true>(w, so, qb - queue_lds, 64u, lane);
                            exact_finish_e<true, ONLY, kChunkE>(g, K, tb, w, bl, 64u, slot0, tg, cap_chunks, result, lane, wflags, probe_bits, sg, in, so, pe);
                            compact(qb);
                        }
                        qbyte = qb;
                    }
                    if (kSubs > 1u) { it0 += (kSubs - 1u) * kEAcc; more = __any(it0 < len); }
                }
            }
        }
        if (qbyte != queue_lds) {  // the home records go with the task: drain
            const uint32_t left = (qbyte - queue_lds) >> 2;
            const ExactRegs g = exact_issue_e<false>(w, so, 0u, left, lane);
            exact_finish_e<false, ONLY, kChunkE>(g, K, tb, w, bl, left, slot0, tg, cap_chunks, result, lane, wflags, probe_bits, sg, in, so, pe);
        }
        if (DIRECT || g_lo + group_waves >= g_hi) break;  // every task was some wave's static first one (small inputs): no round trip to the counter for nothing
        uint32_t nxt_task = 0;
        if (lane == 0) nxt_task = atomicAdd(ctr, 1u);  // (drawn only now: a wave that reserved its next task early would hold it hostage at the end of the launch)
        t = g_lo + group_waves + __builtin_amdgcn_readfirstlane(nxt_task);
    }
    emit_epilogue(bl, hole_list + blockIdx.x, tg);
    if (DIRECT || STAGE) {  // what the waves still hold goes out together: one returning atomic per block
        const uint32_t mine = __builtin_amdgcn_readfirstlane(sg.n);
        if (lane == 0u) stg.wave_n[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0u) {
            uint32_t total = 0;
            for (int k = 0; k < WAVES; k++) total += stg.wave_n[k];
            stg.base = total ? atomicAdd(&result[2], (unsigned long long)total) : 0ull;
        }
        __syncthreads();
        unsigned long long pos = stg.base;
        for (uint32_t k = 0; k < wave; k++) pos += stg.wave_n[k];
        if (STAGE) stage_copy_g(sg.buf, mine, pos, tg, lane);
        else stage_copy_e(sg.buf, mine, pos, tg, result, lane);
    }
}

// single-pass emit + hole fix-up: leaves result[0] = number of pairs, out[0..P) contiguous
bool launch_emit_e(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof, bool contacts_only,
                   bool skip_deferred, bool res_filter) {
    EmitTarget tg{out, capacity, ws.scratch, ws.scratch_cap, ws.defer_list, ws.defer_cap};
    const uint32_t tasks = (in.n + 63u) / 64u;
    // Fewer tasks than the chip has wave slots: every task is shared out over four waves by window kind (eight for the smallest inputs, two
    // per kind set), and below 320 tasks the blocks shrink to four waves so that the few tasks reach more CUs.  Thresholds from size sweeps
    // of this kernel (tests/microbench/ab_r3ae.sh .. ab_r3ag.sh, profiles/r03_small_inputs.txt): a task is a chain of dependent round trips,
    // and an input that does not fill the chip has nothing else to hide it behind.
    // The 4-wave kernels are the hole-free ones (DIRECT); a pack that small keeps the 12-wave sequence with its fix-up, which is what publishes
    // the count its members' lists are split by on the device.
    const bool shared = tasks < 4608u, narrow = tasks < 320u && !in.per_model, eight = narrow && tasks < 160u;  // (eight: 6bft's 128 tasks run 3 us faster that way, an S2 cloud of the same size 1.3 us slower)
    const bool direct = narrow;
    const uint32_t split = eight ? 8u : (shared ? 4u : 1u);
    const uint32_t per = narrow ? 4u : (uint32_t)kEWaves, cap = narrow ? 1536u : kEBlocks, want = (split * tasks + per - 1u) / per;
    const uint32_t nb = want < 1 ? 1 : (want > cap ? cap : want);
    if (prof) prof->begin("pairs_emit", st);
#define ARP_LAUNCH_E(W, S, O, ...) hipLaunchKernelGGL((k_emit<W, S, O, ##__VA_ARGS__>), dim3(nb), dim3(W * 64), 0, st, in, (const GridParams *)ws.grid, (const DevParams *)ws.params, \
                                                 (const uint32_t *)ws.cell_start, ws.sorted, tg, ws.hole_list, ws.task_ctr, ws.result)
    // res_filter: the residue-rule kernels (the engine asks for them when the input's residues are runs of atoms, and had k_place write the residue
    // words); the 4-wave kernels of the smallest inputs have no such variant (a call of that size is launches and round trips, not batches)
    // stage: the hole-free sequence of the task-split 12-wave kernels (k_emit<.., STAGE>: records staged per wave in the scratch block, no fix-up launch) -- for
    // inputs the engine's memo says defer nothing (skip_deferred: the kernel raises status bit 128 if they do after all and the host repeats the call with the
    // chunked sequence), whose waves' regions fit the scratch block
    // (up to kStageTasks tasks = 131 k atoms: every record crosses the L2 twice more on this route, which costs the kernel ~1 us per 3 x 10^5 records -- S2 per step,
    // staged against chunks + fix-up: 3 x 10^4 atoms 53 against 64 us, 6 x 10^4 64 / 71, 10^5 76 / 82 (S1 65 / 80), 1.9 x 10^5 106 / 102, 2.9 x 10^5 146 / 129)
    constexpr uint32_t kStageTasks = 2048;
    const bool stage = shared && !narrow && tasks <= kStageTasks && !in.per_model && skip_deferred && (unsigned long long)nb * kEWaves * kWaveStageRecords <= ws.scratch_cap;
    if (stage) {
        if (contacts_only) { if (res_filter) ARP_LAUNCH_E(kEWaves, 4, true, false, true, true); else ARP_LAUNCH_E(kEWaves, 4, true, false, false, true); }
        else { if (res_filter) ARP_LAUNCH_E(kEWaves, 4, false, false, true, true); else ARP_LAUNCH_E(kEWaves, 4, false, false, false, true); }
    } else if (contacts_only) {
        if (eight) ARP_LAUNCH_E(4, 8, true, true);
        else if (narrow) ARP_LAUNCH_E(4, 4, true, true);
        else if (shared) { if (res_filter) ARP_LAUNCH_E(kEWaves, 4, true, false, true); else ARP_LAUNCH_E(kEWaves, 4, true); }
        else { if (res_filter) ARP_LAUNCH_E(kEWaves, 1, true, false, true); else ARP_LAUNCH_E(kEWaves, 1, true); }
    } else {
        if (eight) ARP_LAUNCH_E(4, 8, false, true);
        else if (narrow) ARP_LAUNCH_E(4, 4, false, true);
        else if (shared) { if (res_filter) ARP_LAUNCH_E(kEWaves, 4, false, false, true); else ARP_LAUNCH_E(kEWaves, 4, false); }
        else { if (res_filter) ARP_LAUNCH_E(kEWaves, 1, false, false, true); else ARP_LAUNCH_E(kEWaves, 1, false); }
    }
#undef ARP_LAUNCH_E
    launch_emit_tail(in, ws, tg, nb, st, prof, skip_deferred, !contacts_only, (direct || stage) ? 1u : (shared ? kSmallChunkRecords : kChunkRecords));
    return direct || stage;  // (the host derives the count and the capacity flag from result[2]: engine.cpp finish_result)
}
static_assert(kEBlocks + 384u <= kMaxHoles && 1536u + 384u <= kMaxHoles, "hole list: one entry per block of either kernel");
