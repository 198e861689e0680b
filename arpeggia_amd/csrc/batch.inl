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

// models per member (max ordinal + 1): one atomic per run of same-owner atoms in a wave (members are contiguous)
__global__ __launch_bounds__(256) void k_pack_models(uint32_t n, uint32_t K, const PackDesc *desc, const uint16_t *model, uint32_t *n_models) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t own = ARP_NONE, m = 0;
    if (i < n) { own = pack_owner(desc, K, i, &PackDesc::first_atom); m = (uint32_t)model[i] + 1u; }
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)own);
    if (__all(own == first || own == ARP_NONE)) {  // the common case: the whole wave belongs to one member
        const uint32_t mx = wave_max_u32(own == ARP_NONE ? 0u : m);
        if (lane == 0 && first != ARP_NONE) atomicMax(&n_models[first], mx);
    } else if (own != ARP_NONE) {
        atomicMax(&n_models[own], m);
    }
}

// exclusive scan of the per-member counts into desc[m].model_off (K <= 65535: one block); total > 65535 models cannot be
// told apart in the 16-bit model field -> status word 1
__global__ __launch_bounds__(1024) void k_pack_scan(uint32_t K, const uint32_t *n_models, PackDesc *desc, uint32_t *status) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (K + 1023u) / 1024u, lo = min(K, threadIdx.x * per), hi = min(K, lo + per);
    uint32_t s = 0;
    for (uint32_t m = lo; m < hi; m++) s += n_models[m];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t t = 0; t < 1024u; t++) { const uint32_t v = part[t]; part[t] = run; run += v; }
        desc[K].model_off = run;
        status[0] = run > 65535u ? 1u : 0u;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t m = lo; m < hi; m++) { desc[m].model_off = run; run += n_models[m]; }
}

// renumber in place: models and residue ids per atom, hydrogen-list offsets and CB / SG atom indices per residue, atom indices per hydrogen
__global__ __launch_bounds__(256) void k_pack_fix(uint32_t n, uint32_t n_res, uint32_t n_h, uint32_t K, const PackDesc *desc, uint16_t *model, uint32_t *res_id,
                                                  uint32_t *res_h_ptr, uint32_t *res_cb, uint32_t *res_sg, uint32_t *res_h_idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const PackDesc d = desc[pack_owner(desc, K, i, &PackDesc::first_atom)];
        model[i] = (uint16_t)(model[i] + d.model_off);
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
// A record's member = the member of its ligand atom i (members are contiguous atom ranges).  The emitter writes a batch's 64 records as one
// run, and a batch comes from one task -- 64 consecutive slots of one model's slab -- so a wave's 64 records almost always share their
// member: one descriptor search for the wave (first lane), a range check for the others, ONE atomic per wave.  (One 64-bit atomic per
// record on a few hundred addresses was 10 ms per launch for a 12 M-record pack: most of the batch path's time with full candidate lists.)
DEVFN uint32_t wave_owner(const PackDesc *desc, uint32_t K, uint32_t i, bool have, bool *uniform) {
    const uint32_t i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);  // (the first ACTIVE lane's: callers keep inactive lanes out by exec)
    const uint32_t m0 = pack_owner(desc, K, i0, &PackDesc::first_atom);
    const uint32_t lo = desc[m0].first_atom, hi = desc[m0 + 1u].first_atom;  // (desc has K + 1 entries: the sentinel holds the totals)
    *uniform = __all(!have || (i >= lo && i < hi));
    return m0;
}
__global__ __launch_bounds__(256) void k_split_count(const unsigned long long *result, const arp_pair *pairs, uint32_t K, const PackDesc *desc, unsigned long long *count) {
    const unsigned long long P = result[0];
    const uint32_t lane = threadIdx.x & 63u;
    for (unsigned long long p0 = ((unsigned long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)); p0 < P; p0 += (unsigned long long)gridDim.x * blockDim.x) {
        const bool have = p0 + lane < P;
        const uint32_t i = pairs[have ? p0 + lane : p0].i;
        bool uniform;
        const uint32_t m0 = wave_owner(desc, K, i, have, &uniform);
        const unsigned long long vm = __ballot(have);  // (by the whole wave, not inside the one-lane branch)
        if (uniform) { if (lane == 0) atomicAdd(&count[m0], (unsigned long long)__popcll(vm)); }
        else if (have) atomicAdd(&count[pack_owner(desc, K, i, &PackDesc::first_atom)], 1ull);
    }
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
// The single-pass emitter's list is in no particular order: every record takes the next free place of its member.  (The ordered
// emitter is not packed: its list is laid out task by task, and a task that straddles two members interleaves their records.)
__global__ __launch_bounds__(256) void k_split_scatter(const unsigned long long *result, const arp_pair *pairs, arp_pair *grouped, uint32_t K, const PackDesc *desc,
                                                       unsigned long long *cursor) {
    const unsigned long long P = result[0];
    const uint32_t lane = threadIdx.x & 63u;
    for (unsigned long long p0 = ((unsigned long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)); p0 < P; p0 += (unsigned long long)gridDim.x * blockDim.x) {
        const bool have = p0 + lane < P;
        arp_pair q = pairs[have ? p0 + lane : p0];
        bool uniform;
        const uint32_t m0 = wave_owner(desc, K, q.i, have, &uniform);
        if (uniform) {  // the wave's records go to one member: one atomic, a coalesced run
            const unsigned long long vm = __ballot(have);
            unsigned long long at = 0;
            if (lane == 0) at = atomicAdd(&cursor[m0], (unsigned long long)__popcll(vm));
            at = wave_first_u64(at);
            const uint32_t base = desc[m0].first_atom;
            q.i -= base; q.j -= base;
            if (have) grouped[at + mbcnt(vm)] = q;
        } else if (have) {
            const uint32_t m = pack_owner(desc, K, q.i, &PackDesc::first_atom), base = desc[m].first_atom;
            q.i -= base; q.j -= base;
            grouped[atomicAdd(&cursor[m], 1ull)] = q;
        }
    }
}

void launch_pack_fix(const PackArrays &pa, hipStream_t st) {
    const uint32_t nb = (std::max(std::max(pa.n, pa.n_res + 1u), pa.n_h) + 255u) / 256u;
    (void)hipMemsetAsync(pa.n_models, 0, sizeof(uint32_t) * pa.K, st);
    hipLaunchKernelGGL(k_pack_models, dim3((pa.n + 255u) / 256u), dim3(256), 0, st, pa.n, pa.K, (const PackDesc *)pa.desc, (const uint16_t *)pa.model, pa.n_models);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, pa.K, (const uint32_t *)pa.n_models, pa.desc, pa.status);
    hipLaunchKernelGGL(k_pack_fix, dim3(nb ? nb : 1u), dim3(256), 0, st, pa.n, pa.n_res, pa.n_h, pa.K, (const PackDesc *)pa.desc, pa.model, pa.res_id, pa.res_h_ptr, pa.res_cb,
                       pa.res_sg, pa.res_h_idx);
}
void launch_pack_split(const PackArrays &pa, const unsigned long long *result, const arp_pair *pairs, arp_pair *grouped, bool ordered, hipStream_t st) {
    (void)hipMemsetAsync(pa.count, 0, sizeof(unsigned long long) * pa.K, st);
    hipLaunchKernelGGL(k_split_count, dim3(1024), dim3(256), 0, st, result, pairs, pa.K, (const PackDesc *)pa.desc, pa.count);
    hipLaunchKernelGGL(k_split_scan, dim3(1), dim3(1024), 0, st, pa.K, (const unsigned long long *)pa.count, pa.offset, pa.cursor);
    (void)ordered;
    hipLaunchKernelGGL(k_split_scatter, dim3(1024), dim3(256), 0, st, result, pairs, grouped, pa.K, (const PackDesc *)pa.desc, pa.cursor);
}
