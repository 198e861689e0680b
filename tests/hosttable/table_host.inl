// The round-1 HOST assembly of the contact table (plane fits, ring rows, row expansion, 10-key sort and sc statistics on the CPU, from
// the GPU's pair list): a second, independent implementation of src/contacts/mod.rs:61-137 that one GPU test compares the device table
// with.  NOT part of libarpeggia_amd.so: table.cpp includes this file only under -DARP_WITH_HOST_TABLE, which arpeggia_amd/build.py sets
// for the test-only library tests/hosttable/build/libarpeggia_amd_hosttable.so (build_host_table_library()).
static arp_status get_contacts_host(arp_context *ctx, arp_structure *s, const char *groups, double vdw_comp, double dist_cutoff, arp_table **out) {
    const bool timing = getenv("ARP_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  get_contacts %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // InteractionComplex::new (complex.rs:36-68)
    arp_atoms view;
    arp_status st = arp_structure_atoms(s, groups, &view);
    if (st != ARP_OK) return st;
    std::vector<PlaneEntry> rings, scp;
    PlaneIndex ring_idx, sc_idx;
    std::vector<int64_t> ring_first, sc_first;  // single-model structures: direct (residue, altloc) -> entry tables
    {
        std::vector<uint8_t> bits;
        std::vector<char> has_ring, has_sc;
        std::vector<Plane> fit_ring, fit_sc;
        plane_atom_bits(*s, &bits, &has_ring, &has_sc);
        fit_planes_host(*s, bits, 1u, &fit_ring);
        fit_planes_host(*s, bits, 2u, &fit_sc);
        build_planes(*s, true, has_ring, &fit_ring, &rings, &ring_idx, &ring_first);
        if (rings.empty()) { set_error("Error building ring positions"); return ARP_ERR_NO_RINGS; }  // complex.rs:50
        build_planes(*s, false, has_sc, &fit_sc, &scp, &sc_idx, &sc_first);
    }
    const bool direct = !sc_first.empty() || s->residues.empty();
    std::unordered_map<std::string, uint32_t> rank;
    for (size_t k = 0; k < s->chain_ids.size(); k++) rank[s->chain_ids[k]] = (uint32_t)k;
    std::vector<char> chain_l(s->chain_ids.size(), 0), chain_r(s->chain_ids.size(), 0);
    for (size_t i = 0; i < s->n; i++) { chain_l[s->chain_rank[i]] = (s->attr[i] & ARP_ATTR_LIGAND) != 0; chain_r[s->chain_rank[i]] = (s->attr[i] & ARP_ATTR_RECEPTOR) != 0; }
    {   // chains without atoms still belong to the sets
        std::vector<std::string> L, R;
        parse_groups(s->chain_ids, groups, &L, &R);
        for (auto &c : L) chain_l[rank[c]] = 1;
        for (auto &c : R) chain_r[rank[c]] = 1;
    }
    for (PlaneEntry &e : rings) { e.chain_rank = rank[e.chain]; e.in_l = chain_l[e.chain_rank]; e.in_r = chain_r[e.chain_rank]; }

    lap("planes + groups");
    // get_atomic_contacts (complex.rs:189-299): the GPU hot path
    arp_params prm;
    arp_default_params(&prm);
    prm.vdw_comp = vdw_comp; prm.dist_cutoff = dist_cutoff;
    prm.flags |= ARP_FLAG_CONTACTS_ONLY;  // only pairs with an interaction become rows: filter them before the copy to the host
    arp_pairs pairs{};
    st = arp_contacts_atomic(ctx, &view, &prm, ARP_MEM_HOST, &pairs);
    if (st != ARP_OK) return st;
    lap("atomic pairs (GPU)");

    // per-atom side-chain plane lookup (the join key of mod.rs:100-110 plus resn, as in collect_sc_stats)
    std::vector<int64_t> atom_sc(s->n, -1);
    {
        // one lookup per (residue, conformer altloc); the residue's atoms pick theirs by altloc
        parallel_for(s->residues.size(), 2048, [&](size_t r0, size_t r1, size_t) {
            std::vector<int64_t> plane_of_alt;
            for (size_t r = r0; r < r1; r++) {
                const ResidueInfo &ri = s->residues[r];
                if (ri.atoms.empty()) continue;
                const uint32_t a0 = ri.atoms[0];
                plane_of_alt.assign(ri.altlocs.size(), -1);
                for (size_t k = 0; k < ri.altlocs.size(); k++) {
                    if (direct) { if (sc_first[r] >= 0) plane_of_alt[k] = sc_first[r] + (int64_t)k; continue; }
                    auto f = sc_idx.find(plane_key(s->model_serial[a0], s->chain.at(a0), s->resi[a0], s->icode.at(a0), ri.altlocs[k].c_str(), s->res_resn.at(a0)));
                    if (f != sc_idx.end()) plane_of_alt[k] = (int64_t)f->second;
                }
                for (uint32_t a : ri.atoms)
                    for (size_t k = 0; k < ri.altlocs.size(); k++)
                        if (strcmp(s->altloc.at(a), ri.altlocs[k].c_str()) == 0) { atom_sc[a] = plane_of_alt[k]; break; }
            }
        });
    }
    auto entity_from_atom = [&](uint32_t a) {  // structs.rs:109-119
        Entity e;
        memcpy(e.chain, s->chain.at(a), 8); memcpy(e.resn, s->res_resn.at(a), 8); memcpy(e.atomn, s->name.at(a), 8);
        memcpy(e.insertion, s->icode.at(a), 4); memcpy(e.altloc, s->altloc.at(a), 4);
        e.resi = s->resi[a]; e.atomi = s->serial[a]; e.atom = (int32_t)a; e.chain_rank = s->chain_rank[a]; e.sc_plane = atom_sc[a];
        return e;
    };
    auto entity_from_ring = [&](const PlaneEntry &r) {  // complex.rs:334-342
        Entity e;
        auto put = [](char *dst, size_t cap, const std::string &v) { memset(dst, 0, cap); memcpy(dst, v.data(), std::min(cap - 1, v.size())); };
        put(e.chain, 8, r.chain); put(e.resn, 8, r.resn); put(e.atomn, 8, "Ring"); put(e.insertion, 4, r.icode); put(e.altloc, 4, r.altloc);
        e.resi = r.resi; e.atomi = 0; e.atom = -1;
        e.chain_rank = r.chain_rank;
        if (direct) e.sc_plane = sc_first[r.res] >= 0 ? sc_first[r.res] + (int64_t)r.alt_k : -1;  // same residue, same conformer altloc
        else {
            auto f = sc_idx.find(plane_key(r.model_serial, r.chain.c_str(), r.resi, r.icode.c_str(), r.altloc.c_str(), r.resn.c_str()));
            e.sc_plane = f == sc_idx.end() ? -1 : (int64_t)f->second;
        }
        return e;
    };
    // one row per set bit: a prefix count gives every worker its own output range (rows keep the pair order)
    std::vector<Row> rows;
    {
        const size_t workers = (size_t)std::max(1, host_threads());
        std::vector<uint64_t> first(workers + 1, 0);
        parallel_for((size_t)pairs.n, 1u << 15, [&](size_t k0, size_t k1, size_t w) {
            uint64_t c = 0;
            for (size_t k = k0; k < k1; k++) c += (uint64_t)__builtin_popcount(pairs.data[k].kind);
            first[w + 1] = c;
        });
        for (size_t w = 0; w < workers; w++) first[w + 1] += first[w];
        rows.reserve(first[workers] + 4 * rings.size() + 64);  // the ring rows are appended later: no regrowth of ~100-byte rows
        rows.resize(first[workers]);
        parallel_for((size_t)pairs.n, 1u << 15, [&](size_t k0, size_t k1, size_t w) {
            Row *out_row = rows.data() + first[w];
            for (size_t k = k0; k < k1; k++) {
                const arp_pair &p = pairs.data[k];
                for (uint32_t bits = p.kind; bits; bits &= bits - 1u)
                    *out_row++ = Row{(uint32_t)s->model_serial[p.i], __builtin_ctz(bits), (double)p.dist, entity_from_atom(p.i), entity_from_atom(p.j)};
            }
        });
    }
    arp_pairs_free(&pairs);
    lap("atom rows");

    // Ring centroids in a hash grid (cell edge = the larger of the two search radii): the reference walks an R*-tree for the
    // ring-atom rows and ALL ring pairs for the ring-ring rows (complex.rs:354-405, O(R^2)); both become linear here.
    const double r2 = dist_cutoff * dist_cutoff;
    const double cell_edge = std::max(std::max(std::fabs(dist_cutoff), 6.0), 1e-3);  // the reference only uses cutoff^2 (complex.rs:303)
    auto cell_of = [&](const double q[3], int64_t c[3]) { for (int k = 0; k < 3; k++) c[k] = (int64_t)std::floor(q[k] / cell_edge); };
    auto cell_key = [](int32_t model, const int64_t c[3]) {
        uint64_t h = (uint64_t)(uint32_t)model * 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < 3; k++) h = (h ^ (uint64_t)c[k]) * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
        return h ^ (h >> 31);
    };
    std::unordered_map<uint64_t, std::vector<uint32_t>> ring_cells;  // key collisions only add candidates: every hit is distance-tested
    bool rings_finite = true;
    for (uint32_t k = 0; k < rings.size(); k++) {
        const PlaneEntry &ring = rings[k];
        if (!ring.has_ord) continue;
        if (!(std::isfinite(ring.plane.c[0]) && std::isfinite(ring.plane.c[1]) && std::isfinite(ring.plane.c[2]))) { rings_finite = false; continue; }
        int64_t c[3];
        cell_of(ring.plane.c, c);
        ring_cells[cell_key(ring.model_serial, c)].push_back(k);
    }
    (void)rings_finite;
    auto for_rings_near = [&](int32_t model, const double q[3], auto &&fn) {
        if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]))) return;
        int64_t c[3], d[3];
        cell_of(q, c);
        for (d[0] = c[0] - 1; d[0] <= c[0] + 1; d[0]++)
            for (d[1] = c[1] - 1; d[1] <= c[1] + 1; d[1]++)
                for (d[2] = c[2] - 1; d[2] <= c[2] + 1; d[2]++) {
                    auto it = ring_cells.find(cell_key(model, d));
                    if (it == ring_cells.end()) continue;
                    for (uint32_t k : it->second) fn(k);
                }
    };
    // get_ring_atom_contacts (complex.rs:301-352) + find_cation_pi (aromatic.rs:14-29)
    {
        // atoms that can ever produce a row are the few positively ionizable ones
        for (size_t a = 0; a < s->n; a++) {
            if (!(s->attr[a] & ARP_ATTR_POS_RESN)) continue;
            const double q[3] = {s->x[a], s->y[a], s->z[a]};
            const ResKey yk{s->model_serial[a], s->chain_rank[a], s->res_ord[a], (s->attr[a] & ARP_ATTR_LIGAND) != 0, (s->attr[a] & ARP_ATTR_RECEPTOR) != 0};
            for_rings_near(s->model_serial[a], q, [&](uint32_t k) {
                const PlaneEntry &ring = rings[k];
                if (ring.model_serial != s->model_serial[a]) return;
                const double dx = q[0] - ring.plane.c[0], dy = q[1] - ring.plane.c[1], dz = q[2] - ring.plane.c[2];
                if (!(dx * dx + dy * dy + dz * dz <= r2)) return;
                const ResKey rk{ring.model_serial, ring.chain_rank, ring.ord, ring.in_l, ring.in_r};
                if (!compare_residues(rk, yk, false)) return;
                const double dist = point_dist(ring.plane, q), theta = point_angle(ring.plane, q);
                if (theta <= 30.0 && dist <= 4.5) rows.push_back(Row{(uint32_t)ring.model_serial, ARP_CationPi, dist, entity_from_ring(ring), entity_from_atom((uint32_t)a)});
            });
        }
    }
    // get_ring_ring_contacts (complex.rs:354-405) + find_pi_pi (aromatic.rs:33-64)
    for (const PlaneEntry &k1 : rings) {
        if (!k1.has_ord || !k1.in_l) continue;
        const ResKey r1{k1.model_serial, k1.chain_rank, k1.ord, k1.in_l, k1.in_r};
        for_rings_near(k1.model_serial, k1.plane.c, [&](uint32_t kk) {
            const PlaneEntry &k2 = rings[kk];
            if (!k2.in_r || k2.model_serial != k1.model_serial) return;
            const double v[3] = {k1.plane.c[0] - k2.plane.c[0], k1.plane.c[1] - k2.plane.c[1], k1.plane.c[2] - k2.plane.c[2]};
            const double dist = norm3(v);
            if (!(dist <= 6.0)) return;
            const ResKey r2k{k2.model_serial, k2.chain_rank, k2.ord, k2.in_l, k2.in_r};
            if (!compare_residues(r1, r2k, true)) return;
            const double theta = point_angle(k1.plane, k2.plane.c), dih = plane_dihedral(k1.plane, k2.plane);
            int code = -1;
            if (dih <= 30.0) { if (theta <= 30.0) code = ARP_PiSandwichStacking; else if (theta <= 60.0) code = ARP_PiDisplacedStacking; else if (theta <= 90.0) code = ARP_PiParallelInPlaneStacking; }
            else if (dih <= 60.0) code = ARP_PiTiltedStacking;
            else if (dih <= 90.0) { if (theta >= 30.0 && theta < 60.0) code = ARP_PiLStacking; else if (dist <= 5.0) code = ARP_PiTStacking; }
            if (code >= 0) rows.push_back(Row{(uint32_t)k1.model_serial, code, dist, entity_from_ring(k1), entity_from_ring(k2)});
        });
    }
    lap("ring rows");
    // sort (mod.rs:120-134): model, from_chain, to_chain, from_resi, from_altloc, from_atomi, to_resi, to_altloc, to_atomi, interaction
    int name_rank[ARP_N_INTERACTIONS];
    {
        std::vector<int> o(ARP_N_INTERACTIONS);
        for (int k = 0; k < ARP_N_INTERACTIONS; k++) o[k] = k;
        std::sort(o.begin(), o.end(), [](int a, int b) { return strcmp(arp_interaction_name(a), arp_interaction_name(b)) < 0; });
        for (int k = 0; k < ARP_N_INTERACTIONS; k++) name_rank[o[k]] = k;
    }
    // Sort keys as plain integers, compared in place (no indirection into the rows): names as big-endian words, so that an
    // unsigned compare is the byte-wise string order polars uses.
    struct SortKey {
        uint32_t model; uint32_t from_chain, to_chain;
        int32_t from_resi; uint32_t from_altloc; int32_t from_atomi, to_resi; uint32_t to_altloc; int32_t to_atomi, interaction;
        uint32_t from_ins, to_ins; double distance; uint32_t idx;
    };
    auto be32 = [](const char *p) { return ((uint32_t)(unsigned char)p[0] << 24) | ((uint32_t)(unsigned char)p[1] << 16) | ((uint32_t)(unsigned char)p[2] << 8) | (uint32_t)(unsigned char)p[3]; };
    std::vector<SortKey> keys(rows.size());
    parallel_for(rows.size(), 1u << 14, [&](size_t k0, size_t k1, size_t) {
    for (size_t k = k0; k < k1; k++) {
        const Row &r = rows[k];
        keys[k] = SortKey{r.model, r.from.chain_rank, r.to.chain_rank, r.from.resi, be32(r.from.altloc), r.from.atomi, r.to.resi, be32(r.to.altloc),
                          r.to.atomi, name_rank[r.interaction], be32(r.from.insertion), be32(r.to.insertion), r.distance, (uint32_t)k};
    }
    });
    auto key_less = [](const SortKey &a, const SortKey &b) {
        if (a.model != b.model) return a.model < b.model;
        if (a.from_chain != b.from_chain) return a.from_chain < b.from_chain;
        if (a.to_chain != b.to_chain) return a.to_chain < b.to_chain;
        if (a.from_resi != b.from_resi) return a.from_resi < b.from_resi;
        if (a.from_altloc != b.from_altloc) return a.from_altloc < b.from_altloc;
        if (a.from_atomi != b.from_atomi) return a.from_atomi < b.from_atomi;
        if (a.to_resi != b.to_resi) return a.to_resi < b.to_resi;
        if (a.to_altloc != b.to_altloc) return a.to_altloc < b.to_altloc;
        if (a.to_atomi != b.to_atomi) return a.to_atomi < b.to_atomi;
        if (a.interaction != b.interaction) return a.interaction < b.interaction;
        // the reference's sort is unstable on full ties; break them deterministically
        if (a.from_ins != b.from_ins) return a.from_ins < b.from_ins;
        if (a.to_ins != b.to_ins) return a.to_ins < b.to_ins;
        if (a.distance != b.distance) return a.distance < b.distance;
        return a.idx < b.idx;
    };
    {   // slices sorted by the workers, then merged pairwise (the order is total, so the result does not depend on the slicing)
        std::vector<size_t> cut{0};
        size_t workers = (size_t)std::max(1, host_threads());
        if (keys.size() / (1u << 14) < workers) workers = std::max<size_t>(1, keys.size() / (1u << 14));
        for (size_t w = 1; w <= workers; w++) cut.push_back(keys.size() * w / workers);
        parallel_for(workers, 1, [&](size_t w0, size_t w1, size_t) {
            for (size_t w = w0; w < w1; w++) std::sort(keys.begin() + cut[w], keys.begin() + cut[w + 1], key_less);
        });
        for (size_t step = 1; step < workers; step *= 2) {
            const size_t n_merges = (workers + 2 * step - 1) / (2 * step);
            parallel_for(n_merges, 1, [&](size_t m0, size_t m1, size_t) {
                for (size_t m = m0; m < m1; m++) {
                    const size_t lo = m * 2 * step, mid = std::min(lo + step, workers), hi = std::min(lo + 2 * step, workers);
                    if (mid < hi) std::inplace_merge(keys.begin() + cut[lo], keys.begin() + cut[mid], keys.begin() + cut[hi], key_less);
                }
            });
        }
    }
    std::vector<uint32_t> order(rows.size());
    for (size_t k = 0; k < rows.size(); k++) order[k] = keys[k].idx;
    lap("sort");
    arp_table *t = new arp_table();
    const size_t n = rows.size();
    t->n = n;
    t->model.resize(n); t->interaction.resize(n); t->from_resi.resize(n); t->from_atomi.resize(n); t->to_resi.resize(n); t->to_atomi.resize(n);
    t->from_atom.resize(n); t->to_atom.resize(n); t->distance.resize(n); t->sc_dist.resize(n); t->sc_dihedral.resize(n); t->sc_angle.resize(n); t->sc_valid.resize(n);
    t->from_chain.resize(n); t->from_resn.resize(n); t->from_atomn.resize(n); t->to_chain.resize(n); t->to_resn.resize(n); t->to_atomn.resize(n);
    t->from_insertion.resize(n); t->from_altloc.resize(n); t->to_insertion.resize(n); t->to_altloc.resize(n);
    parallel_for(n, 1u << 14, [&](size_t k_begin, size_t k_end, size_t) {
    uint64_t last_pair = ~0ull;
    float last_sc[3] = {0.f, 0.f, 0.f};
    for (size_t k = k_begin; k < k_end; k++) {
        const Row &r = rows[order[k]];
        t->model[k] = r.model; t->interaction[k] = r.interaction; t->distance[k] = (float)r.distance;  // mod.rs:148
        t->from_chain.set(k, r.from.chain); t->from_resn.set(k, r.from.resn); t->from_atomn.set(k, r.from.atomn);
        t->from_insertion.set(k, r.from.insertion); t->from_altloc.set(k, r.from.altloc);
        t->from_resi[k] = r.from.resi; t->from_atomi[k] = r.from.atomi; t->from_atom[k] = r.from.atom;
        t->to_chain.set(k, r.to.chain); t->to_resn.set(k, r.to.resn); t->to_atomn.set(k, r.to.atomn);
        t->to_insertion.set(k, r.to.insertion); t->to_altloc.set(k, r.to.altloc);
        t->to_resi[k] = r.to.resi; t->to_atomi[k] = r.to.atomi; t->to_atom[k] = r.to.atom;
        // collect_sc_stats (complex.rs:137-174): res1 = ligand residue, res2 = receptor residue
        if (r.from.sc_plane >= 0 && r.to.sc_plane >= 0) {  // rows are sorted by residue pair: consecutive rows mostly share the planes
            const uint64_t pk = ((uint64_t)r.from.sc_plane << 32) | (uint64_t)r.to.sc_plane;
            if (pk != last_pair) {
                const Plane &p1 = scp[r.from.sc_plane].plane, &p2 = scp[r.to.sc_plane].plane;
                last_sc[0] = (float)point_dist(p1, p2.c); last_sc[1] = (float)plane_dihedral(p1, p2); last_sc[2] = (float)point_angle(p1, p2.c);
                last_pair = pk;
            }
            t->sc_valid[k] = 1;
            t->sc_dist[k] = last_sc[0]; t->sc_dihedral[k] = last_sc[1]; t->sc_angle[k] = last_sc[2];
        }
    }
    });
    lap("columns + sc stats");
    *out = t;
    return ARP_OK;
}
