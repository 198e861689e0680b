// Pair search, classification and emission kernels + the launch sequence.  Included by kernels.hip inside namespace arp.
//
// One wave-task = 64 consecutive slots of the cell-sorted order; lane = home atom (kept in registers).
// Half shell: the rest of the home cell and its kx neighbours towards +x, the 2 kx + 1 cells of row (y+1, z) around the home
// column and the three such runs of layer z+1 -- five contiguous slot windows per lane because cells are x-major (and kx
// times finer along x than the cutoff, GridParams::kx).  Every unordered pair is tested
// exactly once; the reference's ordered pair (x in L, y in R) is recovered by candidate(), of which at most one
// orientation can hold (complex.rs:108-130).
//
// For each of the five window kinds the wave stages the covering slot interval through a private LDS buffer in
// chunks (coalesced 16-byte loads); every lane then walks the part of ITS window inside the chunk with one
// ds_read_b128 per test.  Phase 1 = f32 distance prefilter, 16 tests per lane pushed into a bit mask; survivors are
// compacted into an LDS queue with wavefront ballots.  Phase 2 runs on full waves of 64 survivors: exact f64
// decision, pair filter, classification.
//
// Three modes share that code:
//   kCountTasks   candidate pairs per wave-task (for the ordered fill, and for sizing an output buffer)
//   kFillOrdered  writes task t's pairs at the scanned offset of task t: output order is a function of the input only
//   kEmit         single pass: classified records are compacted per wave and flushed as 64-record units (1 KiB
//                 coalesced stores) into chunks that a block-level LDS bump allocator carves out of ONE global
//                 counter (one device atomic per 4096 records).  The few unused chunk tails ("holes", one per
//                 block) are closed afterwards by k_fixup, which moves the tail of the array into them.
enum PairMode { kCountTasks = 0, kFillOrdered = 1, kEmit = 2, kCountContacts = 3 };
//   kCountContacts  kCountTasks for ARP_FLAG_CONTACTS_ONLY: classifies, counts only the pairs with an interaction

constexpr int kWavesPerBlock = 8;
constexpr int kQueue = 128;
constexpr uint32_t kChunk = 256;            // neighbour records per staged chunk (4 KB)
constexpr uint32_t kBlock = 16;             // prefilter tests per lane between two compaction steps
constexpr uint32_t kReadAhead = 4;         // LDS reads in flight per lane in the prefilter (more costs a wave of occupancy in registers)
constexpr uint32_t kPairBlocks = 256 * 8;   // ordered modes: blocks, each owning a contiguous range of wave-tasks
constexpr int kEmitWavesPerSimd = 6;       // register budget of the emit kernel: 80 VGPRs at 6, 64 at 8
constexpr uint32_t kEmitBlocks = (1024u * kEmitWavesPerSimd) / kWavesPerBlock;  // emit mode: blocks of 8 waves, 6 waves per SIMD
constexpr uint32_t kGrab = 1;              // wave-tasks drawn per atomic
constexpr uint32_t kESlotBits = 24;        // k_emit (pairs_emit.inl): neighbour slot bits of a queue entry (v_mul_u32_u24 turns the entry into the record offset); launch_emit routes larger inputs to k_pairs
// records per global allocation (one device atomic each).  The wave whose allocation crosses the end of the block's chunk fetches the
// next one while the block's other waves sleep: 2048 -> 4096 halves those stalls (emit 221 -> 208 us); 8192 gains 2 us more and costs
// the fix-up 8 us (holes grow with the chunk).
constexpr uint32_t kChunkRecords = 4096, kSmallChunkRecords = 2048;  // records per allocation chunk (powers of two; the smaller one: k_emit's four-way task split; its 4-wave kernels stage and flush instead)

template <int MODE>
struct WaveLds {                                  // per-wave LDS working set
    float4 nrec[kChunk + kBlock];                 // f32 prefilter records of the staged chunk (+ kBlock: over-reads stay in bounds)
    uint2 queue[kQueue];                          // phase-1 survivors: (home slot, neighbour slot)
};
struct BlockLds {
    unsigned long long alloc_state;               // emit mode: current chunk index << 32 | records handed out of it
    unsigned long long defer_state;               // the same for the deferred-probe list (kDeferChunk entries per chunk)
    uint32_t chunk_shift;                         // log2 of the records per chunk of alloc_state (one value per launch sequence: k_fixup is told)
};
constexpr uint32_t chunk_shift_of(uint32_t chunk) { uint32_t s = 0; while ((1u << s) < chunk) s++; return s; }
constexpr uint32_t kDeferChunk = 512;          // deferred-list entries per global allocation
constexpr unsigned long long kAllocEmpty = 0xFFFFFFFFull << 32;  // no chunk yet: | chunk size = "exhausted"

// mask = 2 * mask + (d2 <= r2f): one compare and one add-with-carry per prefilter test
DEVFN void push_pass(uint32_t &mask, float d2, float thr) {
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(d2), "v"(thr) : "vcc");
}

// One compaction round in 8 vector instructions (hipcc's version of the same loop body takes 16-18: it carries the queue length
// in a vector register and rebuilds the lane mask from a 0/1 value).  Every lane with a surviving test (mask != 0) appends
// {home slot, neighbour slot of its highest set bit} to the LDS queue at byte address qaddr + 8 * (its rank among those lanes)
// and clears the bit.  Returns the ballot of the lanes that appended.
DEVFN unsigned long long compact_round(uint32_t &mask, uint32_t home_slot, uint32_t nb_minus, uint32_t qaddr) {
    unsigned long long m, save;
    uint32_t t, lz, slot, bm;
    asm volatile(
        "v_cmp_ne_u32 vcc, 0, %[mask]\n\t"
        "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
        "v_ffbh_u32 %[lz], %[mask]\n\t"
        "v_lshl_add_u32 %[t], %[t], 3, %[qaddr]\n\t"
        "v_add_u32 %[slot], %[nb], %[lz]\n\t"
        "v_lshrrev_b32 %[bm], %[lz], %[top]\n\t"
        "s_mov_b64 %[m], vcc\n\t"
        "s_and_saveexec_b64 %[save], vcc\n\t"
        "ds_write2_b32 %[t], %[home], %[slot] offset1:1\n\t"
        "v_xor_b32 %[mask], %[mask], %[bm]\n\t"
        "s_mov_b64 exec, %[save]"
        : [mask] "+v"(mask), [t] "=&v"(t), [lz] "=&v"(lz), [slot] "=&v"(slot), [bm] "=&v"(bm), [m] "=&s"(m), [save] "=&s"(save)
        : [home] "v"(home_slot), [nb] "v"(nb_minus), [qaddr] "s"(qaddr), [top] "s"(0x80000000u)
        : "vcc", "memory");
    return m;
}

struct EmitTarget {  // positions >= capacity spill into the engine's scratch so that a buffer of exactly P records suffices
    arp_pair *out; unsigned long long capacity;
    arp_pair *scratch; unsigned long long scratch_cap;
    uint2 *defer_list; unsigned long long defer_cap;  // candidates whose classification needs a hydrogen / disulfide probe
};
// Record stores are non-temporal: streamed past the L2 instead of left dirty in it.  The list is write-once and 16 B x P (460 MB on
// the headline input); kept in the L2 it evicts the sorted records the gathers want and its write-back lands on whatever runs next
// (measured: emit 244 -> 222 us, the following grid build 113 -> 88 us).
typedef uint32_t rec_u32x4 __attribute__((ext_vector_type(4)));
DEVFN void store_record(uint4 *p, const uint4 &r) {
    const rec_u32x4 v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<rec_u32x4 *>(p));
}
DEVFN uint4 *emit_slot(const EmitTarget &tg, unsigned long long pos, unsigned long long *result) {
    if (pos < tg.capacity) return reinterpret_cast<uint4 *>(tg.out) + pos;
    const unsigned long long q = pos - tg.capacity;
    if (q < tg.scratch_cap) return reinterpret_cast<uint4 *>(tg.scratch) + q;
    atomicOr(&result[1], 1ull);
    return nullptr;
}

// Output slots for the n (<= 64) valid records of a batch.  Called by a whole wave; lane 0 talks to the block's LDS bump
// allocator (chunk index << 32 | records used) and, once per chunk, to the global chunk counter.  The one allocation that
// crosses the end of the chunk takes the rest of it (n0 records at pos0), fetches the next chunk and continues there
// (pos1); waves that arrive while it does so sleep on the LDS word until the new chunk is published.
struct Slots { unsigned long long pos0, pos1; uint32_t n0; };
DEVFN unsigned long long wave_first_u64(unsigned long long v) {
    // (the builtin returns int: the casts keep the low word from being sign-extended over the high one -- harmless while positions stay below 2^31, wrong beyond)
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}
// Only the LDS atomic itself runs in lane 0; everything derived from its (broadcast) result is wave-uniform, so the
// position arithmetic lands on the scalar unit instead of costing vector issue slots for one live lane.
// The LDS atomic is a round trip of several hundred cycles under load, in the middle of a batch's dependent chain: alloc_issue starts it
// (lane 0) as soon as the record count is known, alloc_finish picks the answer up after the classification has been computed.
DEVFN unsigned long long alloc_issue(unsigned long long &state, uint32_t n, uint32_t lane) {
    unsigned long long old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&state, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return old;  // (lane 0 holds the answer)
}
// (chunk = records per chunk, a power of two given by its log2: a constant in the hot kernel, read from the block's LDS in the others)
DEVFN Slots alloc_finish_rt(unsigned long long &state, unsigned long long *g_head, uint32_t n, uint32_t lane, unsigned long long old, uint32_t shift) {
    const uint32_t CHUNK = 1u << shift;
    for (;;) {
        old = wave_first_u64(old);
        const uint32_t used = (uint32_t)old, chunk = (uint32_t)(old >> 32);
        if (used + n <= CHUNK) return Slots{((unsigned long long)chunk << shift) + used, 0ull, n};
        if (used <= CHUNK) {  // this allocation crosses the end: it alone refills
            const uint32_t n0 = CHUNK - used;
            unsigned long long nc = 0;
            if (lane == 0) {
                nc = atomicAdd(g_head, 1ull);
                __hip_atomic_store(&state, (nc << 32) | (unsigned long long)(n - n0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            nc = wave_first_u64(nc);
            return Slots{((unsigned long long)chunk << shift) + used, nc << shift, n0};
        }
        // exhausted while another wave refills: sleep on the LDS word until the new chunk is published, then retry
        while ((uint32_t)(wave_first_u64(__hip_atomic_load(&state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 32) == chunk)
            __builtin_amdgcn_s_sleep(2);
        old = alloc_issue(state, n, lane);
    }
}
template <uint32_t CHUNK>
DEVFN Slots alloc_finish(unsigned long long &state, unsigned long long *g_head, uint32_t n, uint32_t lane, unsigned long long old) {
    static_assert((CHUNK & (CHUNK - 1u)) == 0u, "chunk sizes are powers of two");
    return alloc_finish_rt(state, g_head, n, lane, old, chunk_shift_of(CHUNK));
}
template <uint32_t CHUNK>
DEVFN Slots alloc_chunked(unsigned long long &state, unsigned long long *g_head, uint32_t n, uint32_t lane) {
    return alloc_finish<CHUNK>(state, g_head, n, lane, alloc_issue(state, n, lane));
}
// The hole-free sequence of small inputs (k_emit's DIRECT kernels stage their records in LDS and take the places of a whole flush at once):
// n places straight from the global counter, which then counts RECORDS -- a few hundred returning atomics per call, all on one address
// (~11 ns each when they queue up), instead of a fix-up launch behind the sequence.
DEVFN Slots alloc_direct(unsigned long long *g_head, uint32_t n, uint32_t lane) {
    unsigned long long p = 0;
    if (lane == 0) p = atomicAdd(g_head, (unsigned long long)n);
    return Slots{wave_first_u64(p), 0ull, n};
}
DEVFN Slots alloc_chunked_rt(unsigned long long &state, unsigned long long *g_head, uint32_t n, uint32_t lane, uint32_t shift) {
    return alloc_finish_rt(state, g_head, n, lane, alloc_issue(state, n, lane), shift);
}

constexpr uint32_t kWaveAllBoth = 1u, kWaveContactsOnly = 2u;  // wave-uniform switches of process_batch, kept in a scalar register
// Record p of the cell-sorted array through a 32-bit byte offset from the (scalar) base: 48 p = (p + 2 p) << 4 is two full-rate
// instructions and the load takes the scalar-base form, where a 64-bit multiply-add is a quarter-rate one.  Valid below 2^32 / 48
// slots; the launchers route larger inputs (kBigSlots) to the variants with inline probes, which address in 64 bits.
constexpr uint32_t kBigSlots = 0x5000000u;
template <bool BIG = false>
DEVFN const Fat &fat_at(const Fat *base, uint32_t p) {
    if (BIG) return base[p];
    uint32_t off;  // (as inline asm: the optimiser folds the shifts back into one quarter-rate v_mul_lo_u32)
    asm("v_lshl_add_u32 %0, %1, 1, %1\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(off) : "v"(p));
    return *reinterpret_cast<const Fat *>(reinterpret_cast<const char *>(base) + (size_t)off);
}
// decision bounds and the pair-rule table -> LDS, once per block (ends with a barrier)
DEVFN void load_lds_params(LdsParams &prm, const DevParams *dprm, const GridParams *gp) {
    const double *src = dprm->s_clash;
    double *dst = prm.s_clash;
    for (uint32_t k = threadIdx.x; k < 3 * 256 + 16; k += blockDim.x) dst[k] = src[k];
    for (uint32_t k = threadIdx.x; k < 512u; k += blockDim.x) prm.lut[k] = pair_lut_entry(k);
    if (threadIdx.x == 0) {
        prm.r2 = dprm->r2; prm.s_ion = dprm->s_ion; prm.s_polar = dprm->s_polar; prm.s_hphob = dprm->s_hphob;
        prm.contacts_only = dprm->flags & ARP_FLAG_CONTACTS_ONLY;
        prm.all_both = gp ? gp->all_both : 0u;
    }
    __syncthreads();
}

// Phase 2 on up to 64 survivors.  Returns the number of valid candidate pairs of the batch.  PROBES == false keeps the
// rare data-dependent rules (hydrogen-bond angle test over the donor residue's hydrogens, disulfide dihedral) out of the
// hot kernel -- they cost it half its occupancy in registers: such pairs go to a list that k_pairs_deferred finishes.
template <int MODE, bool PROBES>
DEVFN uint32_t process_batch(const DevAtoms &in, const LdsParams &prm, const Sorted &so, WaveLds<MODE> &w, BlockLds &bl, uint2 ent, bool active,
                             unsigned long long base, uint32_t emitted, const EmitTarget &tg, unsigned long long *result, uint32_t lane, uint32_t wflags) {
    bool valid = false, swap = false;
    double s = 0.0;
    Fat a, b;
    const bool all_both = !PROBES && (wflags & kWaveAllBoth);  // wave-uniform (held in a scalar register): a scalar branch, no divergence
    if (active) {
        a = fat_at<PROBES>(so.fat, ent.x); b = fat_at<PROBES>(so.fat, ent.y);  // the probe variants keep 64-bit addressing (inputs of any size)
        s = sq_dist(a.x, a.y, a.z, b.x, b.y, b.z);
        const int o = all_both ? orient_all_both(a, b) : orient(a, b);
        valid = (s <= prm.r2) & (o != 0);  // rstar: inclusive
        swap = o == 2;
    }
    unsigned long long vm = __ballot(valid);
    const uint32_t nvalid = (uint32_t)__popcll(vm);
    if (MODE != kCountTasks) {
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (valid) {
            if (PROBES) r.w = classify<true>(in, prm, s, a, b, swap, result);
            else r.w = classify_fast(prm, s, a.pw, b.pw, in.n_res != 0u ? 1u : 0u);
            r.x = swap ? b.orig : a.orig; r.y = swap ? a.orig : b.orig;
            r.z = __float_as_uint(dist_f32(s));
        }
        if (wflags & kWaveContactsOnly) {  // ARP_FLAG_CONTACTS_ONLY: candidates without any interaction are dropped (kDeferKind != 0 stays)
            valid = valid && r.w != 0u;
            vm = __ballot(valid);
        }
        if (MODE == kCountContacts) return (uint32_t)__popcll(vm);
        if (MODE == kFillOrdered) {
            const unsigned long long pos = base + emitted + mbcnt(vm);
            if (valid && pos < tg.capacity) reinterpret_cast<uint4 *>(tg.out)[pos] = r;
            if (!PROBES) {
                // Fast ordered fill: a record whose rules need a probe is written with a placeholder kind at its final position
                // and listed as {entry, position} -- two list slots, allocated in even counts so that a pair never straddles a
                // chunk -- for k_patch_deferred, which overwrites the kind in place.
                const bool defer = valid && r.w == kDeferKind && pos < tg.capacity;
                const unsigned long long dm = __ballot(defer);
                if (dm) {
                    const Slots ds = alloc_chunked<kDeferChunk>(bl.defer_state, &result[3], 2u * (uint32_t)__popcll(dm), lane);
                    if (defer) {
                        const uint32_t dr = 2u * mbcnt(dm);
                        const unsigned long long p = dr < ds.n0 ? ds.pos0 + dr : ds.pos1 + (dr - ds.n0);
                        if (p + 1ull < tg.defer_cap) { tg.defer_list[p] = ent; tg.defer_list[p + 1ull] = make_uint2((uint32_t)pos, (uint32_t)(pos >> 32)); }
                        else atomicOr(&result[1], 8ull);
                    }
                }
            }
        } else {
            if (!PROBES) {
                const bool defer = valid && r.w == kDeferKind;
                const unsigned long long dm = __ballot(defer);
                if (dm) {  // hand the candidates to the deferred pass.  The list is carved in kDeferChunk-entry chunks by the
                           // same block-level allocator as the records: one device atomic per wave and batch on result[3]
                           // serialised hydrogen-rich inputs at ~90 ns each (measured: 2.0 ms on a 150k-atom structure).
                    const Slots ds = alloc_chunked<kDeferChunk>(bl.defer_state, &result[3], (uint32_t)__popcll(dm), lane);
                    if (defer) {
                        const uint32_t dr = mbcnt(dm);
                        const unsigned long long p = dr < ds.n0 ? ds.pos0 + dr : ds.pos1 + (dr - ds.n0);
                        if (p < tg.defer_cap) tg.defer_list[p] = ent; else atomicOr(&result[1], 8ull);
                    }
                    valid = valid && !defer;
                    vm = __ballot(valid);
                }
            }
            const uint32_t n = (uint32_t)__popcll(vm);
            if (n) {  // compacted, coalesced store of the batch's records straight from registers
                const Slots sl = alloc_chunked_rt(bl.alloc_state, &result[2], n, lane, bl.chunk_shift);
                const uint32_t rank = mbcnt(vm);
                if (sl.n0 == n && sl.pos0 + n <= tg.capacity) {
                    // the common case, decided on the scalar unit: one run inside the caller's buffer -> scalar base + 32-bit lane offset
                    uint4 *run = reinterpret_cast<uint4 *>(tg.out) + sl.pos0;
                    if (valid) store_record(&run[rank], r);
                } else if (valid) {  // the run crosses a chunk end or the end of the caller's buffer (scratch until k_fixup)
                    uint4 *d = emit_slot(tg, rank < sl.n0 ? sl.pos0 + rank : sl.pos1 + (rank - sl.n0), result);
                    if (d) store_record(d, r);
                }
            }
        }
        return (uint32_t)__popcll(vm);
    }
    return nvalid;
}

// Block epilogue of the emit mode: the unused tail of the block's last chunk is its (one) hole.  Every wave must call it.
DEVFN void emit_epilogue(BlockLds &bl, ulonglong2 *hole, const EmitTarget &tg) {
    __syncthreads();
    {   // unused tail of the block's last deferred-list chunk: sentinels the deferred passes skip
        const unsigned long long st = bl.defer_state;
        const uint32_t chunk = (uint32_t)(st >> 32), used = (uint32_t)st;
        if (chunk != 0xFFFFFFFFu)
            for (uint32_t k = used + threadIdx.x; k < kDeferChunk; k += blockDim.x) {
                const unsigned long long p = (unsigned long long)chunk * kDeferChunk + k;
                if (p < tg.defer_cap) tg.defer_list[p] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            }
    }
    if (threadIdx.x == 0) {
        const unsigned long long st = bl.alloc_state;
        const uint32_t chunk = (uint32_t)(st >> 32), used = (uint32_t)st;
        unsigned long long hs = 0, hl = 0;
        const uint32_t per = 1u << bl.chunk_shift;
        if (chunk != 0xFFFFFFFFu && used < per) { hs = ((unsigned long long)chunk << bl.chunk_shift) + used; hl = per - used; }
        *hole = make_ulonglong2(hs, hl);
    }
}

template <int MODE, bool PROBES>
__global__ __launch_bounds__(kWavesPerBlock * 64, ((MODE == kEmit || MODE == kFillOrdered) && !PROBES) ? kEmitWavesPerSimd : 1) void k_pairs(DevAtoms in, const GridParams *gp, const DevParams *dprm, const uint32_t *cell_start,
                                                               Sorted so, uint32_t *task_count, const unsigned long long *task_base,
                                                               EmitTarget tg, ulonglong2 *hole_list, uint32_t *task_ctr, unsigned long long *result) {
    __shared__ LdsParams prm;
    __shared__ WaveLds<MODE> wl[kWavesPerBlock];
    __shared__ BlockLds bl;
    load_lds_params(prm, dprm, gp);
    if (threadIdx.x == 0) {
        bl.alloc_state = kAllocEmpty | kChunkRecords;  // "exhausted": the first allocation fetches a chunk
        bl.defer_state = kAllocEmpty | kDeferChunk;
        bl.chunk_shift = chunk_shift_of(kChunkRecords);
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nx = gp->nx, ny = gp->ny, nzt = gp->nzt, kx = gp->kx, n_heavy = gp->n_heavy, n_tasks = gp->n_tasks;
    const uint32_t sy = gp->sy_shift;
    const uint32_t wflags = (gp->all_both ? kWaveAllBoth : 0u) | ((dprm->flags & ARP_FLAG_CONTACTS_ONLY) ? kWaveContactsOnly : 0u);
    const double r2m = gp->r2m;
    WaveLds<MODE> &w = wl[wave];
    // Task distribution: blocks b and b+8 share an XCD (and its private L2), so block group (b mod 8) owns one contiguous
    // eighth of the task range -- the windows its waves stage come out of that L2 -- and inside a group the waves draw
    // kGrab tasks at a time from the group's counter, which evens out the very different costs of surface and core tasks.
    const uint32_t n_groups = min(8u, gridDim.x);
    const uint32_t group = blockIdx.x % n_groups;
    const uint32_t g_lo = (uint32_t)(((unsigned long long)n_tasks * group) / n_groups), g_hi = (uint32_t)(((unsigned long long)n_tasks * (group + 1u)) / n_groups);
    uint32_t *ctr = task_ctr + (MODE * 8 + group) * kTaskCtrStride;
    uint32_t qlen = 0;   // phase-1 survivors waiting in w.queue (wave-uniform); emit mode carries them across tasks
    // LDS byte address of this wave's queue (for compact_round)
    const uint32_t queue_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint2 *)w.queue);
    // The first task of every wave is static (its index among the group's waves): 6144 waves hitting the counters at
    // launch would be handed their first task one by one.  Later tasks come from the counter, which starts behind them.
    const uint32_t group_waves = ((gridDim.x - group + n_groups - 1u) / n_groups) * kWavesPerBlock;
    uint32_t t0 = g_lo + ((blockIdx.x / n_groups) * kWavesPerBlock + wave) * kGrab;
#pragma unroll 1
    for (;; ) {
    if (t0 >= g_hi) break;  // (moving on to the next group's counter instead of retiring was measured: +12 % -- the loop-variant range costs registers)
    const uint32_t t1 = min(t0 + kGrab, g_hi);
#pragma unroll 1
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t a = t * 64u + lane;  // this lane's home slot
        const bool have = a < n_heavy;
        float4 home = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t cx = 0, cy = 0, cz = 0;
        if (have) {
            home = so.rec[a];
            uint32_t c = so.fat[a].cell;
            const uint32_t row = c / nx;
            cx = c - row * nx; grid_row_decode(row, ny, nzt, sy, cy, cz);
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
        const unsigned long long base = (MODE == kFillOrdered) ? task_base[t] : 0ull;
        // per-lane constants of the prefilter: -2 h (exact in f32) and the threshold r2m - |h|^2, rounded up
        const float3 hm2 = make_float3(-2.0f * home.x, -2.0f * home.y, -2.0f * home.z);
        const float thr = __double2float_ru(r2m - ((double)home.x * home.x + (double)home.y * home.y + (double)home.z * home.z));
        uint32_t emitted = 0;
#pragma unroll 1
        for (int k = 0; k < 5; k++) {
            uint32_t lo = wlo[0], hi = whi[0];
#pragma unroll
            for (int j = 1; j < 5; j++) if (k == j) { lo = wlo[j]; hi = whi[j]; }
            const bool nonempty = lo < hi;
            const uint32_t L = wave_min_u32(nonempty ? lo : 0xFFFFFFFFu), H = wave_max_u32(nonempty ? hi : 0u);
            if (L >= H) continue;
#pragma unroll 1
            for (uint32_t cs = L; cs < H; cs += kChunk) {
                const uint32_t ce = min(cs + kChunk, H);
                const uint32_t j0 = max(lo, cs), j1 = min(hi, ce);
                const uint32_t len = (nonempty && j1 > j0) ? j1 - j0 : 0u;
                if (!__any(len != 0u)) {  // no lane's window reaches into this chunk: on to the first slot any lane still needs (lanes of one task can sit in
                    // rows whose neighbour rows lie far apart in the cell order -- a whole y strip apart at a strip's edge, arp_internal.h grid_row)
                    const uint32_t need = wave_min_u32((nonempty && hi > ce) ? max(lo, ce) : 0xFFFFFFFFu);
                    if (need >= H) break;
                    cs = need - kChunk;  // (>= cs: the chunk was a whole one, or no lane would be left; the loop's increment follows)
                    continue;
                }
                wave_lds_fence();  // previous chunk fully consumed
                for (uint32_t p = cs + lane; p < ce; p += 64u) w.nrec[p - cs] = so.rec[p];
                wave_lds_fence();
                const uint32_t off = len ? j0 - cs : 0u;
#pragma unroll 1
                for (uint32_t it0 = 0; __any(it0 < len); it0 += kBlock) {
                    // Phase 1: kBlock prefilter tests per lane, results pushed into a per-lane bit mask (test u -> bit kBlock-1-u).
                    // Lanes whose window is exhausted read slots 0..kBlock-1 (any staged data will do: their bits are dropped).
                    const uint32_t wbase = it0 < len ? off + it0 : 0u;
                    const float4 *win = w.nrec + wbase;
                    uint32_t mask = 0;
                    // kReadAhead LDS reads in flight, then their tests: issued one by one, every test would pay the full LDS latency
#pragma unroll
                    for (uint32_t u0 = 0; u0 < kBlock; u0 += kReadAhead) {
                        float rx[kReadAhead], ry[kReadAhead], rz[kReadAhead], rw[kReadAhead];
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) { const float4 r = win[u0 + u]; rx[u] = r.x; ry[u] = r.y; rz[u] = r.z; rw[u] = r.w; }
                        // |n|^2 - 2 n.h against thr = r2m - |h|^2: 5 VALU per test.  The three FMAs of ONE test are a dependent chain (~2 ns per
                        // link on a SIMD, tests/microbench/valu_rate.hip); issued test-major they serialise, so the loop runs link-major over the
                        // kReadAhead tests in flight: neighbouring instructions are independent.
                        float acc[kReadAhead];
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(rx[u], hm2.x, rw[u]);
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(ry[u], hm2.y, acc[u]);
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) acc[u] = __fmaf_rn(rz[u], hm2.z, acc[u]);
#pragma unroll
                        for (uint32_t u = 0; u < kReadAhead; ++u) push_pass(mask, acc[u], thr);
                    }
                    const uint32_t rem = len > it0 ? len - it0 : 0u;  // tests past the window end read other atoms: drop them
                    if (rem < kBlock) mask &= ~((1u << (kBlock - rem)) - 1u);
                    // Compaction: one round per surviving test of the busiest lane; every round appends <= 64 entries
                    // (mask bit 31 - lz <-> test u = lz - (32 - kBlock) <-> neighbour slot cs + wbase + u)
                    const uint32_t nb_minus = cs + wbase - (32u - kBlock);
                    while (__any(mask != 0u)) {
                        const uint32_t q0 = __builtin_amdgcn_readfirstlane(qlen);
                        const unsigned long long m = compact_round(mask, a, nb_minus, queue_lds + 8u * q0);
                        uint32_t q1 = q0 + (uint32_t)__popcll(m);
                        if (q1 >= 64) {
                            q1 -= 64;
                            wave_lds_fence();  // lanes read entries other lanes wrote
                            const uint2 ent = w.queue[q1 + lane];
                            wave_lds_fence();
                            emitted += process_batch<MODE, PROBES>(in, prm, so, w, bl, ent, true, base, emitted, tg, result, lane, wflags);
                        }
                        qlen = q1;
                    }
                }
            }
        }
        // Survivors left at the task boundary.  The ordered modes keep their bookkeeping per task, so they drain now.
        if (MODE != kEmit && qlen) {
            const bool act = lane < qlen;
            wave_lds_fence();
            const uint2 ent = act ? w.queue[lane] : make_uint2(0u, 0u);
            wave_lds_fence();
            emitted += process_batch<MODE, PROBES>(in, prm, so, w, bl, ent, act, base, emitted, tg, result, lane, wflags);
            qlen = 0;
        }
        if ((MODE == kCountTasks || MODE == kCountContacts) && lane == 0) task_count[t] = emitted;
    }
    uint32_t nxt = 0;
    if (lane == 0) nxt = atomicAdd(ctr, kGrab);  // (drawn only now: a wave that reserved its next task early would hold it hostage at the end of the launch)
    t0 = g_lo + group_waves * kGrab + __builtin_amdgcn_readfirstlane(nxt);
    }
    if (MODE == kFillOrdered && !PROBES) emit_epilogue(bl, hole_list + blockIdx.x, tg);  // (its hole entry is unused: sentinels only)
    if (MODE == kEmit) {
        if (qlen) {
            const bool act = lane < qlen;
            wave_lds_fence();
            const uint2 ent = act ? w.queue[lane] : make_uint2(0u, 0u);
            wave_lds_fence();
            process_batch<MODE, PROBES>(in, prm, so, w, bl, ent, act, 0ull, 0u, tg, result, lane, wflags);
        }
        emit_epilogue(bl, hole_list + blockIdx.x, tg);
    }
}

// The deferred pass of the emit mode: candidates that need a hydrogen or disulfide probe, classified with the probes
// inline and emitted through the same allocator (its blocks add their own holes to the list k_fixup closes).
constexpr uint32_t kDeferBlocks = 384;       // fills the chip at this kernel's 3 waves per SIMD when the list is long
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_pairs_deferred(DevAtoms in, const DevParams *dprm, Sorted so, EmitTarget tg,
                                                                         ulonglong2 *hole_list, unsigned long long *result, uint32_t chunk_shift) {
    __shared__ LdsParams prm;
    __shared__ WaveLds<kEmit> wl[kWavesPerBlock];
    __shared__ BlockLds bl;
    const unsigned long long n = min(result[3] * kDeferChunk, tg.defer_cap);  // result[3] counts list chunks
    if (n == 0ull) {  // nothing was deferred (no hydrogens, no close CYS SG pair): no parameter tables, no hole
        if (threadIdx.x == 0) hole_list[blockIdx.x] = make_ulonglong2(0ull, 0ull);
        return;
    }
    load_lds_params(prm, dprm, nullptr);
    if (threadIdx.x == 0) {
        bl.alloc_state = kAllocEmpty | (1u << chunk_shift);  // (the chunk size of the emit kernel that ran before: one counter, one unit)
        bl.defer_state = kAllocEmpty | kDeferChunk;
        bl.chunk_shift = chunk_shift;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wflags = (dprm->flags & ARP_FLAG_CONTACTS_ONLY) ? kWaveContactsOnly : 0u;
    for (unsigned long long e0 = ((unsigned long long)blockIdx.x * kWavesPerBlock + wave) * 64ull; e0 < n; e0 += (unsigned long long)gridDim.x * kWavesPerBlock * 64ull) {
        uint2 ent = make_uint2(0xFFFFFFFFu, 0u);
        if (e0 + lane < n) ent = tg.defer_list[e0 + lane];
        const bool act = ent.x != 0xFFFFFFFFu;  // chunk tails hold sentinels
        process_batch<kEmit, true>(in, prm, so, wl[wave], bl, ent, act, 0ull, 0u, tg, result, lane, wflags);
    }
    emit_epilogue(bl, hole_list + blockIdx.x, tg);
}

// The deferred pass of the fast ordered fill and of the single-pass emitter's all-candidates mode: {entry, position} pairs; the probes
// decide the kind, which is patched in place (positions beyond the caller's capacity live in the engine's scratch until k_fixup).
__global__ __launch_bounds__(kWavesPerBlock * 64) void k_patch_deferred(DevAtoms in, const DevParams *dprm, Sorted so, EmitTarget tg,
                                                                         unsigned long long *result) {
    __shared__ LdsParams prm;
    const unsigned long long n = min(result[3] * kDeferChunk, tg.defer_cap) / 2ull;  // {entry, position} pairs
    if (n == 0ull) return;  // nothing was deferred (no hydrogens, no close CYS SG pair)
    load_lds_params(prm, dprm, nullptr);
    const uint4 *list = reinterpret_cast<const uint4 *>(tg.defer_list);
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (unsigned long long)gridDim.x * blockDim.x) {
        const uint4 e = list[q];
        if (e.x == 0xFFFFFFFFu) continue;  // chunk tail
        const Fat a = so.fat[e.x], b = so.fat[e.y];
        const double s = sq_dist(a.x, a.y, a.z, b.x, b.y, b.z);
        const bool swap = orient(a, b) == 2;
        const uint32_t kind = classify<true>(in, prm, s, a, b, swap, result);
        const unsigned long long pos = ((unsigned long long)e.w << 32) | e.z;
        uint4 *d = emit_slot(tg, pos, result);
        if (d) reinterpret_cast<uint32_t *>(d)[3] = kind;
    }
}

// Close the holes of the emit pass: with R = records reserved and P = R - sum(holes) valid ones, every hole slot below P
// is filled with a valid record from [P, R).  No sorting is needed: a hole is the tail of ONE 2048-record chunk, so the
// valid records above P are described by a per-chunk table over the <= n_holes + 1 chunks that [P, R) spans, and the
// hole slots below P by the holes themselves in any order.  Each block rebuilds that (tiny) plan in LDS -- two
// 2048-wide scans, two entries per thread -- and then takes part in a grid-stride copy.
constexpr uint32_t kFixThreads = 1024;
constexpr uint32_t kMaxHoles = 2 * kFixThreads;  // the launches keep emit blocks + deferred blocks < kMaxHoles
DEVFN unsigned long long scan1024_u64(unsigned long long v, unsigned long long *red, unsigned long long *total) {
    const uint32_t i = threadIdx.x;
    unsigned long long inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(inc, off);
        if ((i & 63) >= (uint32_t)off) inc += t;
    }
    __syncthreads();
    if ((i & 63) == 63) red[i >> 6] = inc;
    __syncthreads();
    unsigned long long before = 0, tot = 0;
    for (uint32_t k = 0; k < kFixThreads / 64; k++) { if (k < (i >> 6)) before += red[k]; tot += red[k]; }
    *total = tot;
    return before + inc - v;  // exclusive
}
// no_deferred_pass: the launcher skipped k_pairs_deferred because the previous call on the same arrays deferred nothing (engine.cpp,
// "deferred-pass memo"); if this call did defer candidates after all, status bit 128 makes the host repeat it with the pass.
__global__ __launch_bounds__(kFixThreads) void k_fixup(const ulonglong2 *hole_list, uint32_t n_holes, const GridParams *g, EmitTarget tg,
                                                       unsigned long long *result, uint32_t no_deferred_pass, uint32_t chunk_shift) {
    __shared__ unsigned long long fstart[kMaxHoles], fpre[kMaxHoles + 1];  // hole parts below P, any order
    __shared__ unsigned long long tpre[kMaxHoles + 1];                     // valid stretch of each chunk of [P, R) ...
    __shared__ unsigned int tlen[kMaxHoles];                               // ... and its length
    __shared__ unsigned long long red[kFixThreads / 64];
    const uint32_t i = threadIdx.x, e0 = 2u * i, e1 = e0 + 1u;
    const ulonglong2 h0 = (e0 < n_holes) ? hole_list[e0] : make_ulonglong2(0ull, 0ull);
    const ulonglong2 h1 = (e1 < n_holes) ? hole_list[e1] : make_ulonglong2(0ull, 0ull);
    const uint32_t sh = chunk_shift;  // records per chunk = 1 << sh (what the emit kernels of this launch sequence allocated in)
    const unsigned long long R = result[2] << sh;
    unsigned long long holes_total;
    scan1024_u64(h0.y + h1.y, red, &holes_total);
    const unsigned long long P = R - holes_total;
    // tail chunks: chunk c0 + e, valid part = [max(chunk base, P), chunk end) minus the chunk's hole (set below)
    const unsigned long long c0 = P >> sh, n_tail = (R >> sh) - c0;  // n_tail <= n_holes + 1 <= kMaxHoles
    auto tail_start = [&](unsigned long long e) { const unsigned long long cb = (c0 + e) << sh; return cb > P ? cb : P; };
    tlen[e0] = (e0 < n_tail) ? (unsigned int)(((c0 + e0 + 1) << sh) - tail_start(e0)) : 0u;
    tlen[e1] = (e1 < n_tail) ? (unsigned int)(((c0 + e1 + 1) << sh) - tail_start(e1)) : 0u;
    __syncthreads();
    auto cut = [&](const ulonglong2 &h) {  // a hole cuts the end off one tail chunk
        if (h.y && (h.x >> sh) >= c0) {
            const unsigned long long j = (h.x >> sh) - c0, ts = tail_start(j);
            tlen[j] = h.x > ts ? (unsigned int)(h.x - ts) : 0u;
        }
    };
    cut(h0); cut(h1);
    __syncthreads();
    auto front = [&](const ulonglong2 &h) { return (h.y && h.x < P) ? ((h.x + h.y < P ? h.x + h.y : P) - h.x) : 0ull; };
    const unsigned long long f0 = front(h0), f1 = front(h1), t0 = tlen[e0], t1 = tlen[e1];
    unsigned long long F, T;
    const unsigned long long fe = scan1024_u64(f0 + f1, red, &F);
    const unsigned long long te = scan1024_u64(t0 + t1, red, &T);
    fstart[e0] = h0.x; fstart[e1] = h1.x;
    fpre[e0] = fe; fpre[e1] = fe + f0;
    tpre[e0] = te; tpre[e1] = te + t0;
    if (i == 0) { fpre[kMaxHoles] = F; tpre[kMaxHoles] = T; }
    __syncthreads();
    if (blockIdx.x == 0 && i == 0) {
        result[0] = P;
        if (P > tg.capacity) result[1] |= 1ull;
        if (F != T) result[1] |= 16ull;  // internal consistency check of the plan
        if (g->bad & 1u) result[1] |= 4ull;
        if (g->bad & 2u) result[1] |= 64ull;
        if (no_deferred_pass && result[3] != 0ull) result[1] |= 128ull;
    }
    if (P > tg.capacity || F != T) return;  // the caller's buffer cannot hold the table: report the size only
    // The copy.  List entry e with a part of f = fpre[e + 1] - fpre[e] slots below P takes the records fpre[e] .. fpre[e] + f of the tail's
    // valid stretches.  A block owns the entries e = blockIdx.x + j * gridDim.x (j < kFixPer: two or three real holes of a 4096-record
    // chunk at most); wave j finds the tail chunk of entry j's first record by ONE binary search (every lane the same LDS words: broadcast
    // reads), then all threads run over the block's slots as ONE flat index space -- every iteration independent, so the loads of a
    // thread's six or so records are in flight together (hole after hole, each copy waited for the previous hole's round trip to memory).
    constexpr uint32_t kFixPer = kMaxHoles / 256u;
    __shared__ unsigned long long b_m0[kFixPer], b_pre[kFixPer + 1], b_dst[kFixPer];
    __shared__ uint32_t b_chunk[kFixPer];
    if (gridDim.x * kFixPer < kMaxHoles) return;  // (launched with 256 blocks)
    const uint32_t wv = i >> 6;
    if (wv < kFixPer) {
        const uint32_t e = blockIdx.x + wv * gridDim.x;
        const unsigned long long m0 = fpre[e];
        uint32_t lo = 0, hi = kMaxHoles;  // last k with tpre[k] <= m0 (zero-length stretches are skipped by taking the last one)
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tpre[mid] <= m0) lo = mid; else hi = mid; }
        if ((i & 63u) == 0u) { b_m0[wv] = m0; b_dst[wv] = fstart[e]; b_chunk[wv] = lo; b_pre[wv + 1] = fpre[e + 1] - m0; }
    }
    __syncthreads();
    if (i == 0) { unsigned long long run = 0; b_pre[0] = 0; for (uint32_t j = 0; j < kFixPer; j++) { run += b_pre[j + 1]; b_pre[j + 1] = run; } }
    __syncthreads();
    const unsigned long long total = b_pre[kFixPer];
    for (unsigned long long q = i; q < total; q += kFixThreads) {
        uint32_t j = 0;
#pragma unroll
        for (uint32_t k = 1; k < kFixPer; k++) j += (b_pre[k] <= q) ? 1u : 0u;  // (b_pre is non-decreasing: the count is the index)
        const unsigned long long r = q - b_pre[j], m = b_m0[j] + r;
        uint32_t c = b_chunk[j];
        while (c + 1u < kMaxHoles && tpre[c + 1u] <= m) c++;
        uint4 *d = emit_slot(tg, b_dst[j] + r, result);
        const uint4 *sp = emit_slot(tg, tail_start(c) + (m - tpre[c]), result);
        if (d && sp) *d = *sp;
    }
}

// ---------------------------------------------------------------------------------------------- profiler + launch
void Profiler::begin(const char *name, hipStream_t st) {
    if (!enabled) return;
    if (!created) { for (int k = 0; k < kMax; k++) { (void)hipEventCreate(&ev0[k]); (void)hipEventCreate(&ev1[k]); } created = true; }
    if (n >= kMax) return;
    names[n] = name;
    (void)hipEventRecord(ev0[n], st);
}
void Profiler::end(hipStream_t st) {
    if (!enabled || n >= kMax) return;
    (void)hipEventRecord(ev1[n], st);
    n++;
}

template <typename TOut, bool ZERO_IN, bool FINISH>
static void launch_scan(uint32_t *in, const uint32_t *n_ptr, TOut *tmp, TOut *out, uint32_t *ticket, const Workspace &ws, unsigned long long capacity,
                        bool have_out, hipStream_t st) {
    (void)ticket;
    hipLaunchKernelGGL(k_scan_reduce<TOut>, dim3(kScanBlocks), dim3(kScanThreads), 0, st, (const uint32_t *)in, n_ptr, tmp);
    hipLaunchKernelGGL((k_scan_apply<TOut, ZERO_IN, FINISH>), dim3(kScanBlocks), dim3(kScanThreads), 0, st, in, n_ptr, (const TOut *)tmp, out,
                       (const GridParams *)ws.grid, ws.result, capacity, have_out ? 1 : 0);
}

static uint32_t blocks_for(uint32_t n, uint32_t cap) {
    uint32_t tasks = (n + 63u) / 64u;
    uint32_t blocks = (tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    return blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
}

unsigned long long emit_scratch_records() { return (unsigned long long)kMaxHoles * kChunkRecords; }

void launch_grid(const DevAtoms &in, const Workspace &ws, hipStream_t st, Profiler *prof, double cutoff, bool ordered, bool want_rkey) {
    const uint32_t n = in.n;
    const uint32_t nb = (n + 255) / 256;
    auto P0 = [&](const char *nm) { if (prof) prof->begin(nm, st); };
    auto P1 = [&]() { if (prof) prof->end(st); };
    if (prof) prof->n = 0;
    // One structure: the blocks of k_cellid size the grid themselves (grid.inl setup_block), and below kCidAllAtoms atoms they also find the box.
    // Packed batches keep the launch of their own (per-model boxes first), and so do the empty input (no k_cellid to fold into) and inputs
    // whose k_cellid blocks no longer fit the chip at once at the folded kernel's occupancy (the launch is noise there).
    const bool fold = n != 0u && !in.per_model && n <= kCidFoldAtoms, all_atoms = fold && n <= kCidAllAtoms;
    P0("grid_bounds");
    const uint32_t nbb = (n + kBoundsThreads - 1u) / kBoundsThreads, bb = nbb < 1 ? 1 : (nbb < kBoundsBlocks ? nbb : kBoundsBlocks);
    if (!all_atoms) hipLaunchKernelGGL(k_bounds, dim3(bb), dim3(kBoundsThreads), 0, st, in, ws.partials);
    if (in.per_model && n) {
        hipLaunchKernelGGL(k_model_box_init, dim3(kPackModels * 6u / 256u), dim3(256), 0, st, ws.model_box);
        hipLaunchKernelGGL(k_model_bounds, dim3(nb), dim3(256), 0, st, in, ws.model_box);
    }
    if (!fold) hipLaunchKernelGGL(k_setup, dim3(1), dim3(256), 0, st, (const double *)ws.partials, bb, ws.grid, ws.params, cutoff, ws.ncells_cap, n, ws.result,
                                  ws.task_ctr, in.per_model ? (const uint32_t *)ws.model_box : (const uint32_t *)nullptr, ws.model_org);
    P1();
    P0("grid_count");
    if (n) {
        const dim3 cb((n + kCidThreads * kCidPer - 1u) / (kCidThreads * kCidPer));
#define ARP_LAUNCH_CID(BOX) hipLaunchKernelGGL(k_cellid<BOX>, cb, dim3(kCidThreads), 0, st, in, ws.grid, ws.params, (const double *)ws.partials, bb, cutoff, ws.ncells_cap, \
                                               ws.result, ws.task_ctr, ws.cell_of_atom, ws.rank_of_atom, ws.cell_count)
        if (all_atoms) ARP_LAUNCH_CID(2); else if (fold) ARP_LAUNCH_CID(1); else ARP_LAUNCH_CID(0);
#undef ARP_LAUNCH_CID
    }
    P1();
    P0("grid_scan");
    if (!in.per_model && n <= kScanOneAtoms) hipLaunchKernelGGL(k_scan_one<true>, dim3(kScanOneBlocks), dim3(kScanOneThreads), 0, st, ws.cell_count, (const uint32_t *)&ws.grid->ncells, ws.result + kScanPartAt, ws.cell_start);
    else hipLaunchKernelGGL(k_scan_single, dim3(in.per_model || n > kCidFoldAtoms ? kScanBlocks : kScanSingleBlocks), dim3(kScanThreads), 0, st, ws.cell_count, (const uint32_t *)&ws.grid->ncells,
                            ws.result + kScanPartAt, ws.cell_start);  // (256 blocks: one round of one word per thread; packs and the largest inputs have tens of MB of cells: 1024)
    P1();
    P0("grid_sort");
    if (ordered) {
        if (n) hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, st, n, (const uint32_t *)ws.cell_of_atom, (const uint32_t *)ws.rank_of_atom,
                                  (const uint32_t *)ws.cell_start, ws.perm, ws.slot_cell);
        hipLaunchKernelGGL(k_gather, dim3(nb ? nb : 1), dim3(256), 0, st, in, ws.grid, (const uint32_t *)ws.cell_start, (const uint32_t *)ws.perm,
                           (const uint32_t *)ws.slot_cell, ws.sorted);
    } else {
        const dim3 pb((n + 256u * kPlacePer - 1u) / (256u * kPlacePer) + (n ? 0u : 1u));
        if (want_rkey) hipLaunchKernelGGL(k_place<true>, pb, dim3(256), 0, st, in, ws.grid, (const uint32_t *)ws.cell_start, (const uint32_t *)ws.cell_of_atom, (const uint32_t *)ws.rank_of_atom, ws.sorted, ws.result);
        else hipLaunchKernelGGL(k_place<false>, pb, dim3(256), 0, st, in, ws.grid, (const uint32_t *)ws.cell_start, (const uint32_t *)ws.cell_of_atom, (const uint32_t *)ws.rank_of_atom, ws.sorted, ws.result);
    }
    P1();
}

// candidate pairs per task + their scan + total (result[0])
void launch_count(const DevAtoms &in, const Workspace &ws, hipStream_t st, Profiler *prof, unsigned long long capacity, bool have_out, bool contacts_only) {
    EmitTarget none{nullptr, 0ull, nullptr, 0ull, nullptr, 0ull};
    if (prof) prof->begin("pairs_count", st);
    auto kern = contacts_only ? k_pairs<kCountContacts, true> : k_pairs<kCountTasks, true>;
    hipLaunchKernelGGL(kern, dim3(blocks_for(in.n, kPairBlocks)), dim3(kWavesPerBlock * 64), 0, st, in, (const GridParams *)ws.grid,
                       (const DevParams *)ws.params, (const uint32_t *)ws.cell_start, ws.sorted, ws.task_count,
                       (const unsigned long long *)ws.task_base, none, ws.hole_list, ws.task_ctr, ws.result);
    if (prof) { prof->end(st); prof->begin("pairs_scan", st); }
    launch_scan<unsigned long long, false, true>(ws.task_count, &ws.grid->n_tasks, ws.scan_tmp64, ws.task_base, ws.tickets + 2, ws, capacity, have_out, st);
    if (prof) prof->end(st);
}

// ordered fill: needs launch_count first
void launch_fill_ordered(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof,
                         bool contacts_only) {
    if (prof) prof->begin("pairs_fill", st);
    if (contacts_only || in.n >= kBigSlots) {
        // the filter needs every kind before a record is placed: probes inline (2 waves per SIMD)
        EmitTarget tg{out, capacity, nullptr, 0ull, nullptr, 0ull};
        hipLaunchKernelGGL((k_pairs<kFillOrdered, true>), dim3(blocks_for(in.n, kPairBlocks)), dim3(kWavesPerBlock * 64), 0, st, in, (const GridParams *)ws.grid,
                           (const DevParams *)ws.params, (const uint32_t *)ws.cell_start, ws.sorted, ws.task_count,
                           (const unsigned long long *)ws.task_base, tg, ws.hole_list, ws.task_ctr, ws.result);
        if (prof) prof->end(st);
        return;
    }
    // every candidate keeps its place: fill at the emit kernel's occupancy, then patch the kinds that needed a probe
    EmitTarget tg{out, capacity, nullptr, 0ull, ws.defer_list, ws.defer_cap};
    hipLaunchKernelGGL((k_pairs<kFillOrdered, false>), dim3(blocks_for(in.n, kEmitBlocks)), dim3(kWavesPerBlock * 64), 0, st, in, (const GridParams *)ws.grid,
                       (const DevParams *)ws.params, (const uint32_t *)ws.cell_start, ws.sorted, ws.task_count,
                       (const unsigned long long *)ws.task_base, tg, ws.hole_list, ws.task_ctr, ws.result);
    if (prof) { prof->end(st); prof->begin("pairs_patch", st); }
    hipLaunchKernelGGL(k_patch_deferred, dim3(kDeferBlocks), dim3(kWavesPerBlock * 64), 0, st, in, (const DevParams *)ws.params, ws.sorted, tg, ws.result);
    if (prof) prof->end(st);
}

// what follows either emit kernel: the deferred probe pass (unless the engine's memo says this input defers nothing) and the hole fix-up
// patch: the emit kernel wrote the deferred candidates as records with a placeholder kind (k_emit, all candidates): k_patch_deferred decides
// the kinds in place; otherwise k_pairs_deferred classifies and emits them itself (its blocks add holes of their own)
// chunk_records == 1: the hole-free sequence of small inputs (k_emit's DIRECT kernels): the records lie back to back from position 0,
// result[2] counts them, the probes ran inline -- nothing follows the emit kernel, and the host derives the count and the status flags
// k_fixup would have published (engine.cpp finish_result).
static void launch_emit_tail(const DevAtoms &in, const Workspace &ws, const EmitTarget &tg, uint32_t nb, hipStream_t st, Profiler *prof, bool skip_deferred, bool patch,
                             uint32_t chunk_records) {
    const uint32_t chunk_shift = chunk_shift_of(chunk_records);
    if (prof) prof->end(st);
    if (chunk_records == 1u) return;
    if (!skip_deferred) {
        if (prof) prof->begin("pairs_deferred", st);
        if (patch) hipLaunchKernelGGL(k_patch_deferred, dim3(kDeferBlocks), dim3(kWavesPerBlock * 64), 0, st, in, (const DevParams *)ws.params, ws.sorted, tg, ws.result);
        else hipLaunchKernelGGL(k_pairs_deferred, dim3(kDeferBlocks), dim3(kWavesPerBlock * 64), 0, st, in, (const DevParams *)ws.params, ws.sorted, tg, ws.hole_list + nb, ws.result, chunk_shift);
        if (prof) prof->end(st);
    }
    if (prof) prof->begin("pairs_fixup", st);
    hipLaunchKernelGGL(k_fixup, dim3(256), dim3(kFixThreads), 0, st, (const ulonglong2 *)ws.hole_list, (skip_deferred || patch) ? nb : nb + kDeferBlocks,
                       (const GridParams *)ws.grid, tg, ws.result, skip_deferred ? 1u : 0u, chunk_shift);
    if (prof) prof->end(st);
}
bool launch_emit_e(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof, bool contacts_only,
                   bool skip_deferred, bool res_filter);
// the residue-rule kernels exist for the inputs launch_emit_e gives a 12-wave kernel: whoever builds the grid asks this before k_place runs
bool emit_takes_res_filter(const DevAtoms &in) {
    const uint32_t tasks = (in.n + 63u) / 64u;
    return in.n < (1u << kESlotBits) - 64u && !(tasks < 320u && !in.per_model);
}
// single-pass emit + hole fix-up: leaves result[0] = number of pairs, out[0..P) contiguous
// Returns true when the hole-free sequence of small inputs ran: result[0] and the flags k_fixup sets are then the host's to derive from
// result[2] (the records) and result[3] (the deferred list's chunks) -- engine.cpp finish_result.
bool launch_emit(const DevAtoms &in, const Workspace &ws, arp_pair *out, unsigned long long capacity, hipStream_t st, Profiler *prof, bool contacts_only,
                 bool skip_deferred, bool res_filter) {
    if (in.n >= kBigSlots) {  // beyond the 32-bit record offsets of the single-pass kernels: count + ordered fill with inline probes
        launch_count(in, ws, st, prof, capacity, true, contacts_only);
        launch_fill_ordered(in, ws, out, capacity, st, prof, contacts_only);
        return false;
    }
    // The default is k_emit (pairs_emit.inl).  This file's k_pairs<kEmit> -- both exact operands gathered, 8-byte queue entries -- is the
    // one alternative kept: it takes the inputs beyond k_emit's 2^24 slots, and arp_debug_set("emit_kernel", 1) selects it for the parity suite.
    const bool gather = g_debug.emit_kernel == 1;  // (arp_debug_set("emit_kernel", 1): the parity suite's run of the alternative kernel)
    if (!gather && in.n < (1u << kESlotBits) - 64u) return launch_emit_e(in, ws, out, capacity, st, prof, contacts_only, skip_deferred, res_filter && emit_takes_res_filter(in));
    EmitTarget tg{out, capacity, ws.scratch, ws.scratch_cap, ws.defer_list, ws.defer_cap};
    const uint32_t nb = blocks_for(in.n, kEmitBlocks);
    if (prof) prof->begin("pairs_emit", st);
    hipLaunchKernelGGL((k_pairs<kEmit, false>), dim3(nb), dim3(kWavesPerBlock * 64), 0, st, in, (const GridParams *)ws.grid, (const DevParams *)ws.params,
                       (const uint32_t *)ws.cell_start, ws.sorted, ws.task_count, (const unsigned long long *)ws.task_base, tg, ws.hole_list,
                       ws.task_ctr, ws.result);
    launch_emit_tail(in, ws, tg, nb, st, prof, skip_deferred, false, kChunkRecords);
    return false;
}
