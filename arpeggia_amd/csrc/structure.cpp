// Structure ingest: the product-side equivalent of the reference's `load_model` (src/utils.rs:51-63) plus the
// per-atom preparation the reference redoes with string matching for every pair (hbond.rs, ionic.rs, hydrophobic.rs):
// here every atom is classified ONCE into an attribute word and the hierarchy is flattened into SoA columns the GPU reads.
#include <algorithm>
#include <array>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_common.h"

namespace arp {

// ---- element classes -------------------------------------------------------------------------------------------
// pdbtbx Element::atomic_radius(): covalent_single = Pyykko & Atsumi 2009, van_der_waals = Alvarez 2013
// (vdw.rs:24-28, hbond.rs:45-52).  C/N/O/S vdW are pinned by the reference's 532-row test; the rest is unpinned.
struct ElemClass { const char *sym; double cov, vdw; };
static const ElemClass kElems[16] = {
    {"C", 0.75, 1.77}, {"N", 0.71, 1.66}, {"O", 0.63, 1.50}, {"S", 1.03, 1.89}, {"H", 0.32, 1.20}, {"P", 1.11, 1.90},
    {"SE", 1.16, 1.82}, {"F", 0.64, 1.46}, {"CL", 0.99, 1.82}, {"BR", 1.14, 1.86}, {"I", 1.33, 2.04}, {"NA", 1.55, 2.50},
    {"MG", 1.39, 2.51}, {"K", 1.96, 2.73}, {"CA", 1.71, 2.62}, {"ZN", 1.18, 2.39}};

static int element_class(const char *sym) {
    for (int k = 0; k < 16; k++) if (strcmp(kElems[k].sym, sym) == 0) return k;
    return -1;
}

// ---- atom classes (data-driven tables) ------------------------------------------------------------------------------
struct ClassSpec { const char *res; const char *atoms; uint32_t bit; };
// keyed by CONFORMER name (hbond.rs:120-121, ionic.rs:45-46)
static const ClassSpec kConformerSpecs[] = {
    // hbond.rs:160-178 donors ("N" of any residue is handled separately)
    {"ARG", "NE NH1 NH2", ARP_ATTR_DONOR}, {"ASN", "ND2", ARP_ATTR_DONOR}, {"GLN", "NE2", ARP_ATTR_DONOR},
    {"HIS", "ND1 NE2", ARP_ATTR_DONOR}, {"LYS", "NZ", ARP_ATTR_DONOR}, {"SER", "OG", ARP_ATTR_DONOR},
    {"THR", "OG1", ARP_ATTR_DONOR}, {"TRP", "NE1", ARP_ATTR_DONOR}, {"TYR", "OH", ARP_ATTR_DONOR}, {"CYS", "SG", ARP_ATTR_DONOR},
    // hbond.rs:137-157 acceptors ("O"/"OXT" of any residue but HOH handled separately)
    {"ASN", "OD1", ARP_ATTR_ACCEPTOR}, {"ASP", "OD1 OD2", ARP_ATTR_ACCEPTOR}, {"GLN", "OE1", ARP_ATTR_ACCEPTOR},
    {"GLU", "OE1 OE2", ARP_ATTR_ACCEPTOR}, {"HIS", "ND1 NE2", ARP_ATTR_ACCEPTOR}, {"SER", "OG", ARP_ATTR_ACCEPTOR},
    {"THR", "OG1", ARP_ATTR_ACCEPTOR}, {"TYR", "OH", ARP_ATTR_ACCEPTOR}, {"MET", "SD", ARP_ATTR_ACCEPTOR}, {"CYS", "SG", ARP_ATTR_ACCEPTOR},
    // ionic.rs:84-99
    {"ARG", "NE CZ NH1 NH2", ARP_ATTR_POS}, {"HIS", "CG ND1 CE1 NE2 CD2", ARP_ATTR_POS}, {"LYS", "NZ", ARP_ATTR_POS},
    {"ASP", "OD1 OD2", ARP_ATTR_NEG}, {"GLU", "OE1 OE2", ARP_ATTR_NEG}};
// keyed by RESIDUE name (hydrophobic.rs:16-17, vdw.rs:50-51, aromatic.rs:18)
static const ClassSpec kResidueSpecs[] = {
    // hydrophobic.rs:27-45 ("CB" of any residue but SER handled separately)
    {"ARG", "CG", ARP_ATTR_HYDROPHOBIC}, {"GLN", "CG", ARP_ATTR_HYDROPHOBIC}, {"GLU", "CG", ARP_ATTR_HYDROPHOBIC}, {"PRO", "CG", ARP_ATTR_HYDROPHOBIC},
    {"ILE", "CG1 CD1 CG2", ARP_ATTR_HYDROPHOBIC}, {"LEU", "CG CD1 CD2", ARP_ATTR_HYDROPHOBIC}, {"LYS", "CG CD", ARP_ATTR_HYDROPHOBIC},
    {"MET", "CG CE SD", ARP_ATTR_HYDROPHOBIC}, {"PHE", "CG CD1 CD2 CE1 CE2 CZ", ARP_ATTR_HYDROPHOBIC}, {"THR", "CG2", ARP_ATTR_HYDROPHOBIC},
    {"TRP", "CG CD2 CE3 CZ3 CH2 CZ2", ARP_ATTR_HYDROPHOBIC}, {"TYR", "CG CD1 CD2 CE1 CE2", ARP_ATTR_HYDROPHOBIC}, {"VAL", "CG1 CG2", ARP_ATTR_HYDROPHOBIC},
    {"CYS", "SG", ARP_ATTR_CYS_SG},
    {"ARG", "NE CZ NH1 NH2", ARP_ATTR_POS_RESN}, {"HIS", "CG ND1 CE1 NE2 CD2", ARP_ATTR_POS_RESN}, {"LYS", "NZ", ARP_ATTR_POS_RESN}};

using ClassMap = std::unordered_map<std::string, uint32_t>;
template <size_t N>
static ClassMap build_map(const ClassSpec (&specs)[N]) {
    ClassMap m;
    for (const ClassSpec &s : specs) {
        std::istringstream is(s.atoms);
        std::string a;
        while (is >> a) m[std::string(s.res) + ":" + a] |= s.bit;
    }
    return m;
}
static uint32_t lookup(const ClassMap &m, const char *res, const char *atom) {
    auto it = m.find(std::string(res) + ":" + atom);
    return it == m.end() ? 0u : it->second;
}

static uint32_t atom_attr(const char *conf_name, const char *res_name, const char *atom, const char *elem, int elem_cls) {
    static const ClassMap by_conf = build_map(kConformerSpecs), by_res = build_map(kResidueSpecs);
    uint32_t a = (uint32_t)elem_cls & ARP_ATTR_ELEM_MASK;
    a |= lookup(by_conf, conf_name, atom) | lookup(by_res, res_name, atom);
    if (strcmp(atom, "N") == 0) a |= ARP_ATTR_DONOR;                                                        // hbond.rs:162-164
    if ((strcmp(atom, "O") == 0 || strcmp(atom, "OXT") == 0) && strcmp(conf_name, "HOH") != 0) a |= ARP_ATTR_ACCEPTOR;  // hbond.rs:139-142
    if (strcmp(elem, "C") == 0 && strcmp(atom, "C") != 0) a |= ARP_ATTR_WEAK_DONOR;                         // hbond.rs:204-207
    if (strcmp(atom, "CB") == 0 && strcmp(res_name, "SER") != 0) a |= ARP_ATTR_HYDROPHOBIC;                 // hydrophobic.rs:29-31
    if (strcmp(elem, "H") == 0) a |= ARP_ATTR_H;
    return a;
}

static bool known_residue(const std::string &upper) {
    static const std::set<std::string> k = {"ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU",
                                            "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL", "HOH"};  // residues.rs:131-161
    return k.count(upper) != 0;
}

// ---- raw records ------------------------------------------------------------------------------------------------
struct Record {
    double x, y, z, occ;
    int32_t serial, resi, model_serial;
    char name[8], resn[8], chain[8], altloc[4], icode[4], elem[4];
    uint32_t res_ord, res_id;  // hierarchy == 1 only
};

static std::string cut(const std::string &line, size_t c0, size_t c1, bool upper) {  // 1-based inclusive columns, trimmed
    if (line.size() < c0) return "";
    std::string s = line.substr(c0 - 1, std::min(c1, line.size()) - (c0 - 1));
    size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
    if (a == std::string::npos) return "";
    s = s.substr(a, b - a + 1);
    if (upper) for (char &ch : s) ch = (char)toupper((unsigned char)ch);
    return s;
}
static void put(char *dst, size_t cap, const std::string &s) {
    memset(dst, 0, cap);
    memcpy(dst, s.data(), std::min(cap - 1, s.size()));
}

static arp_status read_pdb(const char *path, std::vector<Record> *out) {
    std::ifstream f(path);
    if (!f) { set_error("cannot open '%s'", path); return ARP_ERR_IO; }
    std::string line;
    int32_t model_serial = 0;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.compare(0, 5, "MODEL") == 0 && (line.size() == 5 || line[5] == ' ')) { model_serial = (int32_t)strtol(cut(line, 7, 14, false).c_str(), nullptr, 10); continue; }
        bool atom = line.compare(0, 6, "ATOM  ") == 0, het = line.compare(0, 6, "HETATM") == 0;
        if ((!atom && !het) || line.size() < 54) continue;
        Record r{};
        r.serial = (int32_t)strtol(cut(line, 7, 11, false).c_str(), nullptr, 10);
        put(r.name, sizeof r.name, cut(line, 13, 16, true));
        put(r.altloc, sizeof r.altloc, cut(line, 17, 17, false));
        put(r.resn, sizeof r.resn, cut(line, 18, 20, true));
        put(r.chain, sizeof r.chain, cut(line, 22, 22, false));
        r.resi = (int32_t)strtol(cut(line, 23, 26, false).c_str(), nullptr, 10);
        put(r.icode, sizeof r.icode, cut(line, 27, 27, false));
        r.x = strtod(cut(line, 31, 38, false).c_str(), nullptr);
        r.y = strtod(cut(line, 39, 46, false).c_str(), nullptr);
        r.z = strtod(cut(line, 47, 54, false).c_str(), nullptr);
        std::string occ = cut(line, 55, 60, false);
        r.occ = occ.empty() ? 1.0 : strtod(occ.c_str(), nullptr);
        std::string el = cut(line, 77, 78, true);
        if (el.empty()) {  // no element column: first letter of the atom name
            const char *p = r.name;
            while (*p && isdigit((unsigned char)*p)) p++;
            el = *p ? std::string(1, *p) : "X";
        }
        put(r.elem, sizeof r.elem, el);
        r.model_serial = model_serial;
        out->push_back(r);
    }
    return ARP_OK;
}

// Minimal mmCIF `_atom_site` loop reader (the reference reads mmCIF through pdbtbx as well, utils.rs:53-57).
static std::vector<std::string> cif_tokens(const std::string &line) {
    std::vector<std::string> t;
    size_t i = 0, n = line.size();
    while (i < n) {
        while (i < n && isspace((unsigned char)line[i])) i++;
        if (i >= n) break;
        if (line[i] == '\'' || line[i] == '"') {
            char q = line[i++];
            size_t j = i;
            while (j < n && !(line[j] == q && (j + 1 == n || isspace((unsigned char)line[j + 1])))) j++;
            t.push_back(line.substr(i, j - i));
            i = j + 1;
        } else {
            size_t j = i;
            while (j < n && !isspace((unsigned char)line[j])) j++;
            t.push_back(line.substr(i, j - i));
            i = j;
        }
    }
    return t;
}
static arp_status read_mmcif(const char *path, std::vector<Record> *out) {
    std::ifstream f(path);
    if (!f) { set_error("cannot open '%s'", path); return ARP_ERR_IO; }
    std::string line;
    std::vector<std::string> cols;
    bool in_loop = false, in_atoms = false;
    std::map<std::string, int> idx;
    auto get = [&](const std::vector<std::string> &t, const char *primary, const char *fallback) -> std::string {
        auto it = idx.find(primary);
        if (it == idx.end() && fallback) it = idx.find(fallback);
        if (it == idx.end() || it->second >= (int)t.size()) return "";
        const std::string &v = t[it->second];
        return (v == "." || v == "?") ? "" : v;
    };
    auto upper = [](std::string s) { for (char &c : s) c = (char)toupper((unsigned char)c); return s; };
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '#') { in_loop = in_atoms = false; cols.clear(); idx.clear(); continue; }
        if (line.compare(0, 5, "loop_") == 0) { in_loop = true; in_atoms = false; cols.clear(); idx.clear(); continue; }
        if (in_loop && line[0] == '_') {
            std::string name = cif_tokens(line)[0];
            if (name.compare(0, 11, "_atom_site.") == 0) { in_atoms = true; idx[name.substr(11)] = (int)cols.size(); }
            cols.push_back(name);
            continue;
        }
        if (!(in_loop && in_atoms)) continue;
        std::vector<std::string> t = cif_tokens(line);
        if (t.size() < cols.size()) continue;
        std::string group = get(t, "group_PDB", nullptr);
        if (group != "ATOM" && group != "HETATM") continue;
        Record r{};
        r.serial = (int32_t)strtol(get(t, "id", nullptr).c_str(), nullptr, 10);
        put(r.name, sizeof r.name, upper(get(t, "auth_atom_id", "label_atom_id")));
        put(r.altloc, sizeof r.altloc, get(t, "label_alt_id", nullptr));
        put(r.resn, sizeof r.resn, upper(get(t, "auth_comp_id", "label_comp_id")));
        put(r.chain, sizeof r.chain, get(t, "auth_asym_id", "label_asym_id"));
        r.resi = (int32_t)strtol(get(t, "auth_seq_id", "label_seq_id").c_str(), nullptr, 10);
        put(r.icode, sizeof r.icode, get(t, "pdbx_PDB_ins_code", nullptr));
        r.x = strtod(get(t, "Cartn_x", nullptr).c_str(), nullptr);
        r.y = strtod(get(t, "Cartn_y", nullptr).c_str(), nullptr);
        r.z = strtod(get(t, "Cartn_z", nullptr).c_str(), nullptr);
        std::string occ = get(t, "occupancy", nullptr);
        r.occ = occ.empty() ? 1.0 : strtod(occ.c_str(), nullptr);
        put(r.elem, sizeof r.elem, upper(get(t, "type_symbol", nullptr)));
        std::string mdl = get(t, "pdbx_PDB_model_num", nullptr);
        r.model_serial = mdl.empty() ? 0 : (int32_t)strtol(mdl.c_str(), nullptr, 10);
        out->push_back(r);
    }
    return ARP_OK;
}

// ---- hierarchy + SoA ---------------------------------------------------------------------------------------------
struct BuildConf { std::string name, altloc; uint32_t ord; };
struct BuildRes {
    uint32_t chain; int32_t resi; std::string icode;
    std::vector<BuildConf> confs;
    std::vector<std::pair<uint32_t, uint32_t>> atoms;  // (conformer ordinal, record index)
    bool keep = true; std::string name;
};

static arp_status build(const std::vector<Record> &recs, int hierarchy, bool drop_zero_occ, arp_structure *s) {
    const size_t nrec = recs.size();
    std::vector<ChainInfo> chains;
    std::vector<BuildRes> res;
    std::vector<uint32_t> rec_res(nrec);
    if (hierarchy == 0) {
        // pdbtbx add_atom: existing chain by id, existing residue by (serial, insertion), existing conformer by (name, altloc)
        std::unordered_map<std::string, uint32_t> chain_of, res_of;
        int32_t cur_serial = 0; int cur_model = -1;
        for (size_t i = 0; i < nrec; i++) {
            const Record &r = recs[i];
            if (cur_model < 0 || r.model_serial != cur_serial) { cur_model++; cur_serial = r.model_serial; }
            std::string ck = std::to_string(cur_model) + "|" + r.chain;
            auto ci = chain_of.find(ck);
            if (ci == chain_of.end()) { ci = chain_of.emplace(ck, (uint32_t)chains.size()).first; chains.push_back({(uint32_t)cur_model, cur_serial, r.chain}); }
            std::string rk = std::to_string(ci->second) + "|" + std::to_string(r.resi) + "|" + r.icode;
            auto ri = res_of.find(rk);
            if (ri == res_of.end()) { ri = res_of.emplace(rk, (uint32_t)res.size()).first; BuildRes b; b.chain = ci->second; b.resi = r.resi; b.icode = r.icode; res.push_back(b); }
            BuildRes &br = res[ri->second];
            uint32_t k = 0;
            for (; k < br.confs.size(); k++) if (br.confs[k].name == r.resn && br.confs[k].altloc == r.altloc) break;
            if (k == br.confs.size()) br.confs.push_back({r.resn, r.altloc, k});
            br.atoms.push_back({k, (uint32_t)i});
            rec_res[i] = ri->second;
        }
        for (BuildRes &b : res) {
            b.name = b.confs[0].name;  // Residue::name(): Some iff all conformers agree
            for (const BuildConf &c : b.confs) if (c.name != b.name) {
                set_error("residue %d%s of chain %s has conformers with different names (the reference panics in load_model)", b.resi, b.icode.c_str(), chains[b.chain].id.c_str());
                return ARP_ERR_BAD_INPUT;
            }
            std::string up = b.name;
            for (char &c : up) c = (char)toupper((unsigned char)c);
            b.keep = known_residue(up);  // utils.rs:60
        }
    } else {
        // flat input: residues are given by res_id, ordinals by res_ord
        std::unordered_map<std::string, uint32_t> chain_of;
        std::unordered_map<uint32_t, uint32_t> res_of;
        int32_t cur_serial = 0; int cur_model = -1;
        for (size_t i = 0; i < nrec; i++) {
            const Record &r = recs[i];
            if (cur_model < 0 || r.model_serial != cur_serial) { cur_model++; cur_serial = r.model_serial; }
            std::string ck = std::to_string(cur_model) + "|" + r.chain;
            auto ci = chain_of.find(ck);
            if (ci == chain_of.end()) { ci = chain_of.emplace(ck, (uint32_t)chains.size()).first; chains.push_back({(uint32_t)cur_model, cur_serial, r.chain}); }
            auto ri = res_of.find(r.res_id);
            if (ri == res_of.end()) { ri = res_of.emplace(r.res_id, (uint32_t)res.size()).first; BuildRes b; b.chain = ci->second; b.resi = r.resi; b.icode = r.icode; b.name = r.resn; res.push_back(b); }
            BuildRes &br = res[ri->second];
            uint32_t k = 0;
            for (; k < br.confs.size(); k++) if (br.confs[k].altloc == r.altloc) break;
            if (k == br.confs.size()) br.confs.push_back({r.resn, r.altloc, k});
            br.atoms.push_back({k, (uint32_t)i});
            rec_res[i] = ri->second;
        }
    }
    // positional index of each surviving residue in its chain; global residue ids
    std::vector<uint32_t> chain_count(chains.size(), 0), new_id(res.size(), ARP_NONE), ord(res.size(), 0);
    uint32_t n_res = 0;
    for (size_t r = 0; r < res.size(); r++) if (res[r].keep) { ord[r] = chain_count[res[r].chain]++; new_id[r] = n_res++; }
    // surviving atoms, input order
    std::vector<uint32_t> keep_idx;
    keep_idx.reserve(nrec);
    for (size_t i = 0; i < nrec; i++) {
        if (!res[rec_res[i]].keep) continue;
        if (drop_zero_occ && recs[i].occ == 0.0) continue;  // python.rs:45-47
        keep_idx.push_back((uint32_t)i);
    }
    const size_t n = keep_idx.size();
    std::vector<uint32_t> new_atom(nrec, ARP_NONE);
    for (size_t k = 0; k < n; k++) new_atom[keep_idx[k]] = (uint32_t)k;
    s->n = n;
    s->x.resize(n); s->y.resize(n); s->z.resize(n); s->occ.resize(n);
    s->serial.resize(n); s->resi.resize(n); s->model_serial.resize(n);
    s->name.resize(n); s->resn.resize(n); s->res_resn.resize(n); s->chain.resize(n); s->altloc.resize(n); s->icode.resize(n); s->elem.resize(n);
    s->res_ord.resize(n); s->res_id.resize(n); s->base_attr.resize(n); s->attr.resize(n); s->chain_rank.resize(n); s->model.resize(n); s->atom_chain.resize(n);
    s->chains = chains;
    std::set<std::string> ids;
    for (const ChainInfo &c : chains) ids.insert(c.id);
    s->chain_ids.assign(ids.begin(), ids.end());  // std::string ordering == Rust &str ordering (byte-wise)
    if (s->chain_ids.size() > 65535 || (chains.empty() ? 0u : chains.back().model_idx) > 65535) { set_error("too many chains / models"); return ARP_ERR_BAD_INPUT; }
    std::unordered_map<std::string, uint16_t> rank;
    for (size_t k = 0; k < s->chain_ids.size(); k++) rank[s->chain_ids[k]] = (uint16_t)k;
    for (size_t k = 0; k < n; k++) {
        const Record &r = recs[keep_idx[k]];
        const BuildRes &br = res[rec_res[keep_idx[k]]];
        s->x[k] = r.x; s->y[k] = r.y; s->z[k] = r.z; s->occ[k] = r.occ;
        s->serial[k] = r.serial; s->resi[k] = r.resi; s->model_serial[k] = r.model_serial;
        s->name.set(k, r.name); s->resn.set(k, r.resn); s->res_resn.set(k, br.name.c_str()); s->chain.set(k, r.chain);
        s->altloc.set(k, r.altloc); s->icode.set(k, r.icode); s->elem.set(k, r.elem);
        s->res_ord[k] = hierarchy ? r.res_ord : ord[rec_res[keep_idx[k]]];
        s->res_id[k] = new_id[rec_res[keep_idx[k]]];
        s->atom_chain[k] = br.chain;
        s->chain_rank[k] = rank[r.chain];
        s->model[k] = (uint16_t)chains[br.chain].model_idx;
        int cls = element_class(r.elem);
        if (cls < 0) { set_error("atom %d (%s %s): element '%s' has no radii in this build", r.serial, r.resn, r.name, r.elem); return ARP_ERR_BAD_INPUT; }
        s->base_attr[k] = atom_attr(r.resn, br.name.c_str(), r.name, r.elem, cls);
    }
    // residue tables in hierarchy order (conformer ordinal, then input order)
    s->residues.assign(n_res, ResidueInfo{});
    s->res_h_ptr.assign(n_res + 1, 0); s->res_cb.assign(n_res, ARP_NONE); s->res_sg.assign(n_res, ARP_NONE);
    for (size_t r = 0; r < res.size(); r++) {
        if (!res[r].keep) continue;
        ResidueInfo &ri = s->residues[new_id[r]];
        ri.chain = res[r].chain; ri.resi = res[r].resi; ri.icode = res[r].icode; ri.name = res[r].name; ri.ord = ord[r];
        std::vector<std::pair<uint32_t, uint32_t>> at = res[r].atoms;
        std::stable_sort(at.begin(), at.end(), [](auto &a, auto &b) { return a.first < b.first; });
        for (auto &pr : at) {
            uint32_t a = new_atom[pr.second];
            if (a == ARP_NONE) continue;
            ri.atoms.push_back(a);
            if (hierarchy) ri.ord = s->res_ord[a];
        }
        for (const BuildConf &c : res[r].confs) ri.altlocs.push_back(c.altloc);
    }
    for (uint32_t r = 0; r < n_res; r++) {
        for (uint32_t a : s->residues[r].atoms) {
            if (s->base_attr[a] & ARP_ATTR_H) s->res_h_idx.push_back(a);
            if (s->res_cb[r] == ARP_NONE && strcmp(s->name.at(a), "CB") == 0) s->res_cb[r] = a;   // vdw.rs:55-58 first CB
            if (s->res_sg[r] == ARP_NONE && strcmp(s->name.at(a), "SG") == 0) s->res_sg[r] = a;   // vdw.rs:59-63 first SG
        }
        s->res_h_ptr[r + 1] = (uint32_t)s->res_h_idx.size();
    }
    s->attr = s->base_attr;
    return ARP_OK;
}

// ---- chain groups (utils.rs:71-115) ---------------------------------------------------------------------------------
arp_status parse_groups(const std::vector<std::string> &all, const char *groups, std::vector<std::string> *lig, std::vector<std::string> *rec) {
    std::string g(groups ? groups : "");
    std::vector<std::string> fields;
    size_t a = 0;
    for (size_t k = 0; k <= g.size(); k++) if (k == g.size() || g[k] == '/') { fields.push_back(g.substr(a, k - a)); a = k + 1; }
    if (fields.size() < 2) { set_error("Invalid chain groups format! Use '/' for all-to-all comparisons."); return ARP_ERR_BAD_GROUPS; }
    auto split = [](const std::string &f) {
        std::set<std::string> out; size_t a = 0;
        for (size_t k = 0; k <= f.size(); k++) if (k == f.size() || f[k] == ',') { if (k > a) out.insert(f.substr(a, k - a)); a = k + 1; }
        return out;
    };
    std::set<std::string> L = split(fields[0]), R = split(fields[1]), A(all.begin(), all.end());
    if (L.empty() && R.empty()) { L = A; R = A; }
    else {
        if (L.empty()) { for (auto &c : A) if (!R.count(c)) L.insert(c); }
        else if (R.empty()) { for (auto &c : A) if (!L.count(c)) R.insert(c); }
        if (L.empty() || R.empty()) { set_error("Empty chain groups!"); return ARP_ERR_EMPTY_GROUPS; }
    }
    lig->assign(L.begin(), L.end()); rec->assign(R.begin(), R.end());
    return ARP_OK;
}

arp_status apply_groups(arp_structure *s, const char *groups) {
    std::vector<std::string> L, R;
    arp_status st = parse_groups(s->chain_ids, groups, &L, &R);
    if (st != ARP_OK) return st;
    std::vector<uint32_t> bits(s->chain_ids.size(), 0);
    for (size_t k = 0; k < s->chain_ids.size(); k++) {
        if (std::binary_search(L.begin(), L.end(), s->chain_ids[k])) bits[k] |= ARP_ATTR_LIGAND;
        if (std::binary_search(R.begin(), R.end(), s->chain_ids[k])) bits[k] |= ARP_ATTR_RECEPTOR;
    }
    for (size_t i = 0; i < s->n; i++) s->attr[i] = s->base_attr[i] | bits[s->chain_rank[i]];
    s->groups_applied = groups; s->groups_valid = true;
    return ARP_OK;
}

}  // namespace arp

using namespace arp;

// ---- C ABI ------------------------------------------------------------------------------------------------------
extern "C" void arp_default_params(arp_params *p) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->vdw_comp = 0.1;     // python.rs:32, cli/contacts.rs:38
    p->dist_cutoff = 6.5;  // python.rs:32, cli/contacts.rs:42
    for (int k = 0; k < 16; k++) { p->cov_radius[k] = kElems[k].cov; p->vdw_radius[k] = kElems[k].vdw; }
    p->h_vdw_radius = 1.20;
}
extern "C" int32_t arp_element_class(const char *symbol) {
    if (!symbol) return -1;
    std::string s(symbol);
    for (char &c : s) c = (char)toupper((unsigned char)c);
    return element_class(s.c_str());
}

extern "C" arp_status arp_structure_load(const char *path, int32_t ignore_zero_occupancy, arp_structure **out) {
    if (!path || !out) { set_error("null argument"); return ARP_ERR_BAD_INPUT; }
    *out = nullptr;
    std::vector<Record> recs;
    std::string p(path), ext;
    size_t dot = p.find_last_of('.');
    if (dot != std::string::npos) ext = p.substr(dot + 1);
    for (char &c : ext) c = (char)tolower((unsigned char)c);
    arp_status st = (ext == "cif" || ext == "mmcif") ? read_mmcif(path, &recs) : read_pdb(path, &recs);
    if (st != ARP_OK) return st;
    arp_structure *s = new arp_structure();
    st = build(recs, 0, ignore_zero_occupancy != 0, s);
    if (st != ARP_OK) { delete s; return st; }
    *out = s;
    return ARP_OK;
}

extern "C" arp_status arp_structure_from_records(const arp_records *rec, int32_t hierarchy, arp_structure **out) {
    if (!rec || !out) { set_error("null argument"); return ARP_ERR_BAD_INPUT; }
    *out = nullptr;
    if (rec->n && (!rec->x || !rec->y || !rec->z || !rec->serial || !rec->resi || !rec->name || !rec->resn || !rec->chain || !rec->element)) {
        set_error("null record column"); return ARP_ERR_BAD_INPUT;
    }
    if (hierarchy && rec->n && (!rec->res_ord || !rec->res_id)) { set_error("hierarchy == 1 needs res_ord and res_id"); return ARP_ERR_BAD_INPUT; }
    std::vector<Record> recs(rec->n);
    auto fixed = [](char *dst, size_t cap, const char *src, size_t w, bool up) {
        memset(dst, 0, cap);
        for (size_t k = 0; k < w && k < cap - 1 && src[k]; k++) dst[k] = up ? (char)toupper((unsigned char)src[k]) : src[k];
    };
    for (uint64_t i = 0; i < rec->n; i++) {
        Record &r = recs[i];
        r.x = rec->x[i]; r.y = rec->y[i]; r.z = rec->z[i]; r.occ = rec->occupancy ? rec->occupancy[i] : 1.0;
        r.serial = rec->serial[i]; r.resi = rec->resi[i]; r.model_serial = rec->model_serial ? rec->model_serial[i] : 0;
        fixed(r.name, sizeof r.name, rec->name + 8 * i, 8, true);
        fixed(r.resn, sizeof r.resn, rec->resn + 8 * i, 8, true);
        fixed(r.chain, sizeof r.chain, rec->chain + 8 * i, 8, false);
        if (rec->altloc) fixed(r.altloc, sizeof r.altloc, rec->altloc + 4 * i, 4, false);
        if (rec->icode) fixed(r.icode, sizeof r.icode, rec->icode + 4 * i, 4, false);
        fixed(r.elem, sizeof r.elem, rec->element + 4 * i, 4, true);
        if (hierarchy) { r.res_ord = rec->res_ord[i]; r.res_id = rec->res_id[i]; }
    }
    arp_structure *s = new arp_structure();
    arp_status st = build(recs, hierarchy ? 1 : 0, false, s);
    if (st != ARP_OK) { delete s; return st; }
    *out = s;
    return ARP_OK;
}

extern "C" void arp_structure_free(arp_structure *s) { delete s; }
extern "C" uint64_t arp_structure_n_atoms(const arp_structure *s) { return s ? s->n : 0; }

extern "C" arp_status arp_structure_atoms(arp_structure *s, const char *groups, arp_atoms *v) {
    if (!s || !v) { set_error("null argument"); return ARP_ERR_BAD_INPUT; }
    arp_status st = apply_groups(s, groups);
    if (st != ARP_OK) return st;
    memset(v, 0, sizeof *v);
    v->n = s->n;
    v->x = s->x.data(); v->y = s->y.data(); v->z = s->z.data();
    v->attr = s->attr.data(); v->res_ord = s->res_ord.data(); v->chain_rank = s->chain_rank.data(); v->model = s->model.data();
    v->res_id = s->res_id.data();
    v->n_res = s->residues.size();
    v->res_h_ptr = s->res_h_ptr.data(); v->res_h_idx = s->res_h_idx.data(); v->res_cb = s->res_cb.data(); v->res_sg = s->res_sg.data();
    v->location = ARP_MEM_HOST;
    return ARP_OK;
}

extern "C" const char *arp_structure_strings(const arp_structure *s, const char *column, int32_t *width) {
    if (!s || !column) return nullptr;
    std::string c(column);
    auto ret = [&](const std::vector<char> &b, int w) { if (width) *width = w; return b.data(); };
    if (c == "chain") return ret(s->chain.buf, 8);
    if (c == "resn") return ret(s->res_resn.buf, 8);
    if (c == "conformer") return ret(s->resn.buf, 8);
    if (c == "atomn") return ret(s->name.buf, 8);
    if (c == "insertion") return ret(s->icode.buf, 4);
    if (c == "altloc") return ret(s->altloc.buf, 4);
    if (c == "element") return ret(s->elem.buf, 4);
    return nullptr;
}
extern "C" const int32_t *arp_structure_ints(const arp_structure *s, const char *column) {
    if (!s || !column) return nullptr;
    std::string c(column);
    if (c == "resi") return s->resi.data();
    if (c == "atomi") return s->serial.data();
    if (c == "model") return s->model_serial.data();
    return nullptr;
}
