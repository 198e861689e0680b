// Device side of packed batches (arp_contacts_atomic_batch).  Included by kernels.hip inside namespace arp.
//
// A pack = the SoA arrays of K independent structures copied back to back, untouched, into one block.  The reference never
// pairs atoms of different models (complex.rs:96-98, 201-207) and every model owns its own z slab of the grid, so giving the
// members disjoint model ordinals makes the pack's pair list exactly the union of the members' lists.  The host only copies
// bytes; renumbering (models, residue ids, atom indices inside the residue tables) and the split of the joint pair list back
// into per-structure lists happen here.

DEVFN uint32_t pack_owner(const PackDesc *desc, uint32_t K, uint32_t v, uint32_t PackDesc::*field) {  // last m with desc[m].field <= v
    uint32_t lo = 0, hi = K;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (desc[mid].*field <= v) lo = mid; else hi = mid; }
    return lo;
}

// (members are contiguous atom ranges: a wave of consecutive atoms, or of records of one emitter batch, is almost always inside one)
struct OwnerCache { uint32_t m = 0, lo = 1, hi = 0; };  // (wave-uniform) the member the wave saw last and its atom range
// member of every lane's atom; lanes without a record get ARP_NONE
DEVFN uint32_t lane_owner(const PackDesc *desc, uint32_t K, uint32_t i, bool have, OwnerCache &c) {
    if (!__any(have)) return ARP_NONE;
    if (!__all(!have || (i >= c.lo && i < c.hi))) {
        const unsigned long long vm = __ballot(have);
        const uint32_t i0 = (uint32_t)__shfl((int)i, (int)__ffsll((long long)vm) - 1);
        c.m = pack_owner(desc, K, i0, &PackDesc::first_atom);
        c.lo = desc[c.m].first_atom; c.hi = desc[c.m + 1u].first_atom;  // (desc has K + 1 entries: the sentinel holds the totals)
    }
    if (!have) return ARP_NONE;
    return (i >= c.lo && i < c.hi) ? c.m : pack_owner(desc, K, i, &PackDesc::first_atom);
}
// models per member (max ordinal + 1): one atomic per run of same-owner atoms in a wave (members are contiguous)
__global__ __launch_bounds__(256) void k_pack_models(uint32_t n, uint32_t K, const PackDesc *desc, const uint32_t *model, uint32_t *n_models) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t own = ARP_NONE, m = 0;
    OwnerCache oc;
    own = lane_owner(desc, K, i, i < n, oc);
    if (i < n) m = min(model[i], 0xFFFFFFFEu) + 1u;
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)own);
    if (__all(own == first || own == ARP_NONE)) {  // the common case: the whole wave belongs to one member
        const uint32_t mx = wave_max_u32(own == ARP_NONE ? 0u : m);
        if (lane == 0 && first != ARP_NONE) atomicMax(&n_models[first], mx);
    } else if (own != ARP_NONE) {
        atomicMax(&n_models[own], m);
    }
}

// exclusive scan of the per-member counts into desc[m].model_off (K <= 65535: one block); a pack of more than 65535 models does not
// fit the per-model boxes of the workspace (Workspace::model_box) -> status word 1 (the members then run one by one)
__global__ __launch_bounds__(1024) void k_pack_scan(uint32_t K, const uint32_t *n_models, PackDesc *desc, uint32_t *status) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (K + 1023u) / 1024u, lo = min(K, threadIdx.x * per), hi = min(K, lo + per);
    uint32_t s = 0;
    for (uint32_t m = lo; m < hi; m++) s = min(s + min(n_models[m], kPackModels), 2u * kPackModels);  // (saturating: a member with 2^32 - 1 models must not wrap the total back into range)
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t t = 0; t < 1024u; t++) { const uint32_t v = part[t]; part[t] = run; run = min(run + v, 2u * kPackModels); }  // (saturating: 1024 parts of up to 2^16 each would still fit, 2^32 - 1 per member would not)
        desc[K].model_off = run;
        status[0] = run > kPackModels - 1u ? 1u : 0u;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t m = lo; m < hi; m++) { desc[m].model_off = run; run += n_models[m]; }
}

// renumber in place: models and residue ids per atom, hydrogen-list offsets and CB / SG atom indices per residue, atom indices per hydrogen.
// A pack whose models do not fit the per-model tables of the workspace (status[0] != 0: more than 65535 in all, or one member with a sparse id
// such as 70 000) is re-run member by member by the host, which only learns of it after the stream has run -- so the kernels that follow must
// stay in bounds whatever the members hold: every atom of such a pack gets model 0 (API v1's 16-bit ordinals could not leave the tables; v2's
// 32-bit ones can -- ADVICE r4).
__global__ __launch_bounds__(256) void k_pack_fix(uint32_t n, uint32_t n_res, uint32_t n_h, uint32_t K, const PackDesc *desc, const uint32_t *status, uint32_t *model,
                                                  uint32_t *res_id, uint32_t *res_h_ptr, uint32_t *res_cb, uint32_t *res_sg, uint32_t *res_h_idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool overfull = status[0] != 0u;  // (written by k_pack_scan, the previous launch on this stream)
    OwnerCache oc;
    const uint32_t own = lane_owner(desc, K, i, i < n, oc);
    if (i < n) {
        const PackDesc d = desc[own];
        model[i] = overfull ? 0u : min(model[i] + d.model_off, kPackModels - 1u);  // (the sum is below kPackModels when the pack is not overfull)
        res_id[i] += d.first_res;
    }
    if (i < n_res) {
        const PackDesc d = desc[pack_owner(desc, K, i, &PackDesc::first_res)];
        res_h_ptr[i] += d.first_h;
        if (res_cb[i] != ARP_NONE) res_cb[i] += d.first_atom;
        if (res_sg[i] != ARP_NONE) res_sg[i] += d.first_atom;
    }
    if (i == n_res) res_h_ptr[n_res] = n_h;
    if (i < n_h) res_h_idx[i] += desc[pack_owner(desc, K, i, &PackDesc::first_h)].first_atom;
}

// ---- split of the joint pair list: pairs per member, offsets, grouped copy with indices rebased to the member ----
// A record's member = the member of its ligand atom i (members are contiguous atom ranges).  Both kernels below give every wave ONE
// CONTIGUOUS range of the list.  The emitter writes a batch's 64 records as one run and a batch comes from one task -- 64 consecutive
// slots of one model's slab -- so a range of a few thousand records mentions a handful of members: a wave adds up its records per member
// first and touches the members' counters once per (wave, member).  Same-address device atomics serialise at ~11 ns each; one per
// 64 records, with a thousand waves on the same three or four members at any moment, made each of these kernels 7-13 ms per
// 14 M-record pack, i.e. nearly all of the batch path's time with full candidate lists.
constexpr uint32_t kSplitBlocks = 1024, kSplitThreads = 256, kSplitSlots = 8;

DEVFN void split_range(unsigned long long P, unsigned long long *lo, unsigned long long *hi) {  // this wave's records
    const unsigned long long waves = (unsigned long long)gridDim.x * (blockDim.x / 64u);
    const unsigned long long per = (((P + waves - 1ull) / waves) + 63ull) & ~63ull;
    const unsigned long long w = (unsigned long long)blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);
    *lo = min(P, w * per); *hi = min(P, *lo + per);
}

// (capacity: the records `pairs` holds.  A pack whose list did not fit -- result[0] > capacity: the fix-up then only reports the size, and the host grows
// the buffer and runs the pack again -- is not split at all: round 5 found the split kernels reading result[0] records out of a buffer that held
// fewer, a fault as soon as the excess left the allocation's padding.)
DEVFN unsigned long long split_records(const unsigned long long *result, unsigned long long capacity) { return result[0] <= capacity ? result[0] : 0ull; }
__global__ __launch_bounds__(kSplitThreads) void k_split_count(const unsigned long long *result, unsigned long long capacity, const arp_pair *pairs, uint32_t K, const PackDesc *desc,
                                                               unsigned long long *count) {
    const unsigned long long P = split_records(result, capacity);
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long lo, hi;
    split_range(P, &lo, &hi);
    OwnerCache oc;
    uint32_t cur = ARP_NONE, cnt = 0;  // (wave-uniform) the run of records of one member being added up
    for (unsigned long long p0 = lo; p0 < hi; p0 += 64u) {
        const bool have = p0 + lane < hi;
        const uint32_t m = lane_owner(desc, K, pairs[have ? p0 + lane : p0].i, have, oc);
        for (unsigned long long rem = __ballot(have); rem;) {  // the distinct members of these 64 records (almost always one)
            const uint32_t mm = (uint32_t)__shfl((int)m, (int)__ffsll((long long)rem) - 1);
            const unsigned long long mask = __ballot(have && m == mm);
            if (mm != cur) {
                if (cnt && lane == 0) atomicAdd(&count[cur], (unsigned long long)cnt);
                cur = mm; cnt = 0;
            }
            cnt += (uint32_t)__popcll(mask);
            rem &= ~mask;
        }
    }
    if (cnt && lane == 0) atomicAdd(&count[cur], (unsigned long long)cnt);
}
// offset[m] = pairs of the members before m, offset[K] = P; cursor[m] = offset[m] (consumed by the scatter)
__global__ __launch_bounds__(1024) void k_split_scan(uint32_t K, const unsigned long long *count, unsigned long long *offset, unsigned long long *cursor) {
    __shared__ unsigned long long part[1024];
    const uint32_t per = (K + 1023u) / 1024u, lo = min(K, threadIdx.x * per), hi = min(K, lo + per);
    unsigned long long s = 0;
    for (uint32_t m = lo; m < hi; m++) s += count[m];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (uint32_t t = 0; t < 1024u; t++) { const unsigned long long v = part[t]; part[t] = run; run += v; }
        offset[K] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint32_t m = lo; m < hi; m++) { offset[m] = run; cursor[m] = run; run += count[m]; }
}
// The single-pass emitter's list is in no particular order: a member's records land in its part of the grouped list in whatever order
// the waves reserve their places.  (The ordered emitter is not packed: its list is laid out task by task, and a task that straddles two
// members interleaves their records.)  A wave first adds up its range per member in a table of kSplitSlots entries, reserves each
// member's places with ONE atomic, then copies; if its range mentions more members than the table holds (packs of tiny structures), the
// rest of the range reserves per 64 records.
struct SplitTable { uint32_t member[kSplitSlots], count[kSplitSlots], run[kSplitSlots]; unsigned long long base[kSplitSlots]; };
__global__ __launch_bounds__(kSplitThreads) void k_split_scatter(const unsigned long long *result, unsigned long long capacity, const arp_pair *pairs, arp_pair *grouped, uint32_t K,
                                                                const PackDesc *desc, unsigned long long *cursor) {
    __shared__ SplitTable tables[kSplitThreads / 64];
    SplitTable &t = tables[threadIdx.x >> 6];  // (private to the wave: its lanes run in lockstep, LDS operations of one wave stay in order)
    const unsigned long long P = split_records(result, capacity);
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long lo, hi;
    split_range(P, &lo, &hi);
    OwnerCache oc;
    uint32_t n_ent = 0;                     // (wave-uniform)
    unsigned long long tabled_end = hi;     // records before this one are in the table
    auto find = [&](uint32_t mm) { uint32_t e = 0; while (e < n_ent && t.member[e] != mm) e++; return e; };
    for (unsigned long long p0 = lo; p0 < hi; p0 += 64u) {
        const bool have = p0 + lane < hi;
        const uint32_t m = lane_owner(desc, K, pairs[have ? p0 + lane : p0].i, have, oc);
        bool fits = true;
        for (unsigned long long rem = __ballot(have); rem;) {  // every member of these 64 records needs an entry before any is counted
            const uint32_t mm = (uint32_t)__shfl((int)m, (int)__ffsll((long long)rem) - 1);
            rem &= ~__ballot(have && m == mm);
            if (find(mm) < n_ent) continue;
            if (n_ent == kSplitSlots) { fits = false; break; }
            if (lane == 0) { t.member[n_ent] = mm; t.count[n_ent] = 0u; }
            __builtin_amdgcn_wave_barrier();
            n_ent++;
        }
        if (!fits) { tabled_end = p0; break; }
        for (unsigned long long rem = __ballot(have); rem;) {
            const uint32_t mm = (uint32_t)__shfl((int)m, (int)__ffsll((long long)rem) - 1);
            const unsigned long long mask = __ballot(have && m == mm);
            const uint32_t e = find(mm);
            if (lane == 0) t.count[e] += (uint32_t)__popcll(mask);
            __builtin_amdgcn_wave_barrier();
            rem &= ~mask;
        }
    }
    if (lane < n_ent) { t.base[lane] = t.count[lane] ? atomicAdd(&cursor[t.member[lane]], (unsigned long long)t.count[lane]) : 0ull; t.run[lane] = 0u; }
    __builtin_amdgcn_wave_barrier();
    for (unsigned long long p0 = lo; p0 < hi; p0 += 64u) {
        const bool have = p0 + lane < hi;
        arp_pair q = pairs[have ? p0 + lane : p0];
        const uint32_t m = lane_owner(desc, K, q.i, have, oc);
        for (unsigned long long rem = __ballot(have); rem;) {
            const uint32_t mm = (uint32_t)__shfl((int)m, (int)__ffsll((long long)rem) - 1);
            const bool mine = have && m == mm;
            const unsigned long long mask = __ballot(mine);
            const uint32_t cnt = (uint32_t)__popcll(mask);
            unsigned long long at;
            if (p0 < tabled_end) {
                const uint32_t e = find(mm);
                at = t.base[e] + t.run[e];
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) t.run[e] += cnt;
                __builtin_amdgcn_wave_barrier();
            } else {  // beyond the table: places reserved per 64 records
                at = 0;
                if (lane == 0) at = atomicAdd(&cursor[mm], (unsigned long long)cnt);
                at = wave_first_u64(at);
            }
            if (mine) {
                const uint32_t base = desc[mm].first_atom;
                q.i -= base; q.j -= base;
                grouped[at + mbcnt(mask)] = q;
            }
            rem &= ~mask;
        }
    }
}

void launch_pack_fix(const PackArrays &pa, hipStream_t st) {
    const uint32_t nb = (std::max(std::max(pa.n, pa.n_res + 1u), pa.n_h) + 255u) / 256u;
    (void)hipMemsetAsync(pa.n_models, 0, sizeof(uint32_t) * pa.K, st);
    hipLaunchKernelGGL(k_pack_models, dim3((pa.n + 255u) / 256u), dim3(256), 0, st, pa.n, pa.K, (const PackDesc *)pa.desc, (const uint32_t *)pa.model, pa.n_models);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, pa.K, (const uint32_t *)pa.n_models, pa.desc, pa.status);
    hipLaunchKernelGGL(k_pack_fix, dim3(nb ? nb : 1u), dim3(256), 0, st, pa.n, pa.n_res, pa.n_h, pa.K, (const PackDesc *)pa.desc, (const uint32_t *)pa.status, pa.model, pa.res_id,
                       pa.res_h_ptr, pa.res_cb, pa.res_sg, pa.res_h_idx);
}
void launch_pack_split(const PackArrays &pa, const unsigned long long *result, const arp_pair *pairs, unsigned long long capacity, arp_pair *grouped, bool ordered, hipStream_t st) {
    (void)hipMemsetAsync(pa.count, 0, sizeof(unsigned long long) * pa.K, st);
    hipLaunchKernelGGL(k_split_count, dim3(kSplitBlocks), dim3(kSplitThreads), 0, st, result, capacity, pairs, pa.K, (const PackDesc *)pa.desc, pa.count);
    hipLaunchKernelGGL(k_split_scan, dim3(1), dim3(1024), 0, st, pa.K, (const unsigned long long *)pa.count, pa.offset, pa.cursor);
    (void)ordered;
    hipLaunchKernelGGL(k_split_scatter, dim3(kSplitBlocks), dim3(kSplitThreads), 0, st, result, capacity, pairs, grouped, pa.K, (const PackDesc *)pa.desc, pa.cursor);
}
