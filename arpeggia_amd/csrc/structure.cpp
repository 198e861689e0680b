// Structure ingest: the product-side equivalent of the reference's `load_model` (src/utils.rs:51-63) plus the
// per-atom preparation the reference redoes with string matching for every pair (hbond.rs, ionic.rs, hydrophobic.rs):
// here every atom is classified ONCE into an attribute word and the hierarchy is flattened into SoA columns the GPU reads.
#include <algorithm>
#include <atomic>
#include <array>
#include <cctype>
#include <chrono>
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

// (residue name, atom name) -> attribute bits.  Keys are the two names packed into one 64-bit word (4 + 4 bytes, NUL padded):
// every name in the tables is at most 3 / 4 characters, so a longer input name simply never matches, as in the reference's
// string `matches!`.
using ClassMap = std::unordered_map<uint64_t, uint32_t>;
static bool pack4(const char *s, uint32_t *out) {
    const size_t n = strlen(s);
    if (n > 4) return false;
    uint32_t v = 0;
    memcpy(&v, s, n);
    *out = v;
    return true;
}
static bool pack_key(const char *res, const char *atom, uint64_t *key) {
    uint32_t r, a;
    if (!pack4(res, &r) || !pack4(atom, &a)) return false;
    *key = ((uint64_t)r << 32) | a;
    return true;
}
template <size_t N>
static ClassMap build_map(const ClassSpec (&specs)[N]) {
    ClassMap m;
    for (const ClassSpec &s : specs) {
        std::istringstream is(s.atoms);
        std::string a;
        while (is >> a) {
            uint64_t key = 0;
            pack_key(s.res, a.c_str(), &key);
            m[key] |= s.bit;
        }
    }
    return m;
}
static uint32_t lookup(const ClassMap &m, const char *res, const char *atom) {
    uint64_t key;
    if (!pack_key(res, atom, &key)) return 0u;
    auto it = m.find(key);
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


// Fixed-column fields straight out of the file buffer (no per-field std::string: a 10k-atom file has 130k fields).
struct Field { const char *b, *e; };  // trimmed [b, e)
static inline Field field(const char *line, size_t len, size_t c0, size_t c1) {  // 1-based inclusive columns
    if (len < c0) return {line, line};
    const char *b = line + (c0 - 1), *e = line + std::min(c1, len);
    while (b < e && (*b == ' ' || *b == '\t')) b++;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t')) e--;
    return {b, e};
}
static inline void put_field(char *dst, size_t cap, Field f, bool upper) {
    memset(dst, 0, cap);
    size_t n = std::min(cap - 1, (size_t)(f.e - f.b));
    for (size_t k = 0; k < n; k++) { const char c = f.b[k]; dst[k] = (upper && c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }  // "C"-locale toupper
}
static inline long field_long(Field f) {  // strtol(base 10) on the trimmed field
    {
        const char *q = f.b;
        bool neg = false;
        if (q < f.e && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
        long v = 0;
        int digits = 0;
        for (; q < f.e && *q >= '0' && *q <= '9' && digits < 18; q++, digits++) v = v * 10 + (*q - '0');
        if (digits < 18) return neg ? -v : v;  // like strtol: stops at the first non-digit, 0 when there is none
    }
    char tmp[24];
    size_t n = std::min(sizeof tmp - 1, (size_t)(f.e - f.b));
    memcpy(tmp, f.b, n); tmp[n] = 0;
    return strtol(tmp, nullptr, 10);
}
static inline double field_double(Field f) {
    // Plain decimals ("-12.345", the only form PDB coordinate columns hold): digits / 10^k with both operands exact doubles is
    // correctly rounded, i.e. the value strtod returns.  Anything else (exponents, inf/nan, > 15 digits) goes to strtod.
    {
        static const double kPow10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
        const char *q = f.b;
        bool neg = false;
        if (q < f.e && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
        uint64_t m = 0;
        int digits = 0, frac = 0;
        bool dot = false, ok = q < f.e;
        for (; q < f.e; q++) {
            if (*q >= '0' && *q <= '9') { m = m * 10 + (uint64_t)(*q - '0'); digits++; if (dot) frac++; }
            else if (*q == '.' && !dot) dot = true;
            else { ok = false; break; }
        }
        if (ok && digits > 0 && digits <= 15) {
            const double v = (double)m / kPow10[frac];
            return neg ? -v : v;
        }
    }
    char tmp[40];
    size_t n = std::min(sizeof tmp - 1, (size_t)(f.e - f.b));
    memcpy(tmp, f.b, n); tmp[n] = 0;
    return strtod(tmp, nullptr);
}

static arp_status read_pdb(const char *path, std::vector<Record> *out) {
    FILE *fp = fopen(path, "rb");
    if (!fp) { set_error("cannot open '%s'", path); return ARP_ERR_IO; }
    std::string buf;
    {
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, fp)) > 0) buf.append(chunk, got);
        fclose(fp);
    }
    out->reserve(buf.size() / 81 + 16);
    int32_t model_serial = 0;
    const char *p = buf.data(), *end = p + buf.size();
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *line = p, *le = nl ? nl : end;
        p = nl ? nl + 1 : end;
        while (le > line && (le[-1] == '\r' || le[-1] == '\n')) le--;
        const size_t len = (size_t)(le - line);
        if (len >= 5 && memcmp(line, "MODEL", 5) == 0 && (len == 5 || line[5] == ' ')) { model_serial = (int32_t)field_long(field(line, len, 7, 14)); continue; }
        if (len < 54) continue;
        const bool atom = memcmp(line, "ATOM  ", 6) == 0, het = memcmp(line, "HETATM", 6) == 0;
        if (!atom && !het) continue;
        out->emplace_back();
        Record &r = out->back();
        memset(&r, 0, sizeof r);
        r.serial = (int32_t)field_long(field(line, len, 7, 11));
        put_field(r.name, sizeof r.name, field(line, len, 13, 16), true);
        put_field(r.altloc, sizeof r.altloc, field(line, len, 17, 17), false);
        put_field(r.resn, sizeof r.resn, field(line, len, 18, 20), true);
        put_field(r.chain, sizeof r.chain, field(line, len, 22, 22), false);
        r.resi = (int32_t)field_long(field(line, len, 23, 26));
        put_field(r.icode, sizeof r.icode, field(line, len, 27, 27), false);
        r.x = field_double(field(line, len, 31, 38));
        r.y = field_double(field(line, len, 39, 46));
        r.z = field_double(field(line, len, 47, 54));
        const Field occ = field(line, len, 55, 60);
        r.occ = occ.b == occ.e ? 1.0 : field_double(occ);
        const Field el = field(line, len, 77, 78);
        if (el.b == el.e) {  // no element column: first letter of the atom name
            const char *q = r.name;
            while (*q && isdigit((unsigned char)*q)) q++;
            r.elem[0] = *q ? *q : 'X';
        } else {
            put_field(r.elem, sizeof r.elem, el, true);
        }
        r.model_serial = model_serial;
    }
    return ARP_OK;
}

// mmCIF `_atom_site` reader (the reference reads mmCIF through pdbtbx as well, utils.rs:53-57).  A real CIF 1.1 lexer over the
// whole file: whitespace-separated tokens, '#' comments, '...' / "..." quoted values (a quote ends only before whitespace),
// semicolon-delimited multi-line text fields, loop_ rows that wrap over lines, and the non-loop key/value form of a one-row category.
// Identity columns as pdbtbx 0.12 picks them (recalled from the crate, not verifiable here -- SURVEY.md Appendix B): atom and
// residue NAMES from label_atom_id / label_comp_id, CHAIN from auth_asym_id and residue NUMBER from auth_seq_id when present (the
// author numbering PDB files carry), else label_asym_id / label_seq_id; altloc label_alt_id; insertion pdbx_PDB_ins_code; model
// pdbx_PDB_model_num.
namespace {
struct CifTok {
    enum Kind { End, Value, Tag, Loop, Data, Other } kind = End;
    const char *b = nullptr, *e = nullptr;
    bool quoted = false;
    bool null_value() const { return !quoted && e - b == 1 && (*b == '.' || *b == '?'); }
};
struct CifLexer {
    const char *p, *end;
    bool bol = true;  // at the beginning of a line
    CifLexer(const char *b, const char *e) : p(b), end(e) {}
    static bool ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
    CifTok next() {
        for (;;) {
            while (p < end && ws(*p)) { bol = (*p == '\n'); p++; }
            if (p >= end) return CifTok{};
            if (*p == '#') { while (p < end && *p != '\n') p++; continue; }
            break;
        }
        CifTok t;
        if (*p == ';' && bol) {  // text field: up to the next line that starts with ';'
            const char *b = p + 1, *q = b;
            for (;;) {
                const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q));
                if (!nl) { q = end; break; }
                if (nl + 1 < end && nl[1] == ';') { q = nl; break; }
                q = nl + 1;
            }
            t.kind = CifTok::Value; t.b = b; t.e = q; t.quoted = true;
            while (t.e > t.b && (t.e[-1] == '\r' || t.e[-1] == '\n')) t.e--;
            p = q < end ? q + 2 : end;  // past "\n;"
            bol = false;
            return t;
        }
        bol = false;
        if (*p == '\'' || *p == '"') {
            const char quote = *p++;
            const char *b = p;
            while (p < end && !(*p == quote && (p + 1 == end || ws(p[1]))) && *p != '\n') p++;
            t.kind = CifTok::Value; t.b = b; t.e = p; t.quoted = true;
            if (p < end && *p == quote) p++;
            return t;
        }
        const char *b = p;
        while (p < end && !ws(*p)) p++;
        t.b = b; t.e = p;
        const size_t len = (size_t)(p - b);
        auto ieq = [&](const char *w, size_t n) { if (len < n) return false; for (size_t k = 0; k < n; k++) if (tolower((unsigned char)b[k]) != w[k]) return false; return true; };
        if (*b == '_') t.kind = CifTok::Tag;
        else if (len == 5 && ieq("loop_", 5)) t.kind = CifTok::Loop;
        else if (ieq("data_", 5)) t.kind = CifTok::Data;
        else if (ieq("save_", 5) || (len == 7 && ieq("global_", 7)) || (len == 5 && ieq("stop_", 5))) t.kind = CifTok::Other;
        else t.kind = CifTok::Value;
        return t;
    }
};
}  // namespace

static arp_status read_mmcif(const char *path, std::vector<Record> *out) {
    FILE *fp = fopen(path, "rb");
    if (!fp) { set_error("cannot open '%s'", path); return ARP_ERR_IO; }
    std::string buf;
    {
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, fp)) > 0) buf.append(chunk, got);
        fclose(fp);
    }
    enum Col { GROUP, ID, LABEL_ATOM, AUTH_ATOM, ALT, LABEL_COMP, AUTH_COMP, LABEL_ASYM, AUTH_ASYM, LABEL_SEQ, AUTH_SEQ, INS, X, Y, Z, OCC, ELEM, MODEL, N_COL };
    static const char *kNames[N_COL] = {"group_pdb", "id", "label_atom_id", "auth_atom_id", "label_alt_id", "label_comp_id", "auth_comp_id", "label_asym_id",
                                        "auth_asym_id", "label_seq_id", "auth_seq_id", "pdbx_pdb_ins_code", "cartn_x", "cartn_y", "cartn_z", "occupancy",
                                        "type_symbol", "pdbx_pdb_model_num"};
    auto col_of = [&](const CifTok &t) -> int {  // _atom_site.<name>, case-insensitive
        static const char pre[] = "_atom_site.";
        const size_t len = (size_t)(t.e - t.b);
        if (len <= 11) return -1;
        for (size_t k = 0; k < 11; k++) if (tolower((unsigned char)t.b[k]) != pre[k]) return -1;
        for (int c = 0; c < N_COL; c++) {
            const size_t n = strlen(kNames[c]);
            if (len - 11 != n) continue;
            bool eq = true;
            for (size_t k = 0; k < n && eq; k++) eq = tolower((unsigned char)t.b[11 + k]) == kNames[c][k];
            if (eq) return c;
        }
        return (int)N_COL;  // an _atom_site column this reader does not use
    };
    auto emit = [&](const CifTok *v) {  // one row: v[c] for the known columns (kind End = absent)
        auto text = [&](int c, int fallback) -> Field {
            const CifTok *t = &v[c];
            if ((t->kind == CifTok::End || t->null_value()) && fallback >= 0) t = &v[fallback];
            if (t->kind == CifTok::End || t->null_value()) return Field{nullptr, nullptr};
            return Field{t->b, t->e};
        };
        const Field group = text(GROUP, -1);
        const size_t gl = (size_t)(group.e - group.b);
        if (!((gl == 4 && memcmp(group.b, "ATOM", 4) == 0) || (gl == 6 && memcmp(group.b, "HETATM", 6) == 0))) return;
        out->emplace_back();
        Record &r = out->back();
        memset(&r, 0, sizeof r);
        auto fld = [](Field f) { return f.b ? f : Field{"", ""}; };
        r.serial = (int32_t)field_long(fld(text(ID, -1)));
        put_field(r.name, sizeof r.name, fld(text(LABEL_ATOM, AUTH_ATOM)), true);
        put_field(r.altloc, sizeof r.altloc, fld(text(ALT, -1)), false);
        put_field(r.resn, sizeof r.resn, fld(text(LABEL_COMP, AUTH_COMP)), true);
        put_field(r.chain, sizeof r.chain, fld(text(AUTH_ASYM, LABEL_ASYM)), false);
        r.resi = (int32_t)field_long(fld(text(AUTH_SEQ, LABEL_SEQ)));
        put_field(r.icode, sizeof r.icode, fld(text(INS, -1)), false);
        r.x = field_double(fld(text(X, -1))); r.y = field_double(fld(text(Y, -1))); r.z = field_double(fld(text(Z, -1)));
        const Field occ = text(OCC, -1);
        r.occ = occ.b ? field_double(occ) : 1.0;
        put_field(r.elem, sizeof r.elem, fld(text(ELEM, -1)), true);
        const Field mdl = text(MODEL, -1);
        r.model_serial = mdl.b ? (int32_t)field_long(mdl) : 0;
    };
    out->reserve(buf.size() / 90 + 16);
    CifLexer lex(buf.data(), buf.data() + buf.size());
    CifTok t = lex.next();
    CifTok single[N_COL + 1];  // the key/value form (a category with one row)
    bool have_single = false;
    while (t.kind != CifTok::End) {
        if (t.kind == CifTok::Loop) {
            std::vector<int> cols;
            bool atoms = false;
            for (t = lex.next(); t.kind == CifTok::Tag; t = lex.next()) { const int c = col_of(t); cols.push_back(c); atoms = atoms || c >= 0; }
            if (cols.empty()) continue;
            CifTok row[N_COL + 1];
            size_t k = 0;
            for (; t.kind == CifTok::Value; t = lex.next()) {
                if (atoms && cols[k] >= 0) row[cols[k]] = t;
                if (++k == cols.size()) {
                    if (atoms) { emit(row); for (CifTok &q : row) q = CifTok{}; }
                    k = 0;
                }
            }
            continue;
        }
        if (t.kind == CifTok::Tag) {
            const int c = col_of(t);
            CifTok v = lex.next();
            if (v.kind == CifTok::Value) { if (c >= 0) { single[c] = v; have_single = true; } t = lex.next(); }
            else t = v;
            continue;
        }
        if (t.kind == CifTok::Data && have_single) { emit(single); for (CifTok &q : single) q = CifTok{}; have_single = false; }
        t = lex.next();
    }
    if (have_single) emit(single);
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
        int32_t cur_serial = 0; int cur_model = -1, prev_model = -1;
        uint32_t prev_chain = 0, prev_res = 0;
        bool have_prev = false, have_res = false;
        for (size_t i = 0; i < nrec; i++) {
            const Record &r = recs[i];
            if (cur_model < 0 || r.model_serial != cur_serial) { cur_model++; cur_serial = r.model_serial; }
            // consecutive records almost always stay in the same chain and residue: the keyed lookups run per change only
            if (!have_prev || cur_model != prev_model || memcmp(r.chain, recs[i - 1].chain, sizeof r.chain) != 0) {
                std::string ck = std::to_string(cur_model) + "|" + r.chain;
                auto ci = chain_of.find(ck);
                if (ci == chain_of.end()) { ci = chain_of.emplace(ck, (uint32_t)chains.size()).first; chains.push_back({(uint32_t)cur_model, cur_serial, r.chain}); }
                prev_chain = ci->second; prev_model = cur_model; have_res = false;
            }
            if (!have_res || r.resi != recs[i - 1].resi || memcmp(r.icode, recs[i - 1].icode, sizeof r.icode) != 0) {
                std::string rk = std::to_string(prev_chain) + "|" + std::to_string(r.resi) + "|" + r.icode;
                auto ri = res_of.find(rk);
                if (ri == res_of.end()) { ri = res_of.emplace(rk, (uint32_t)res.size()).first; BuildRes b; b.chain = prev_chain; b.resi = r.resi; b.icode = r.icode; res.push_back(b); }
                prev_res = ri->second; have_res = true;
            }
            have_prev = true;
            BuildRes &br = res[prev_res];
            uint32_t k = 0;
            for (; k < br.confs.size(); k++) if (br.confs[k].name == r.resn && br.confs[k].altloc == r.altloc) break;
            if (k == br.confs.size()) br.confs.push_back({r.resn, r.altloc, k});
            br.atoms.push_back({k, (uint32_t)i});
            rec_res[i] = prev_res;
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
    // (API v2: chain ranks and model ordinals are 32-bit -- the reference keys on the chain id STRING, complex.rs:19-21, and has no chain limit)
    if (s->chain_ids.size() >= 0xFFFFFFF0ull || chains.size() >= 0xFFFFFFF0ull) { set_error("too many chains / models"); return ARP_ERR_BAD_INPUT; }
    std::unordered_map<std::string, uint32_t> rank;
    for (size_t k = 0; k < s->chain_ids.size(); k++) rank[s->chain_ids[k]] = (uint32_t)k;
    std::vector<uint32_t> rank_of_chain(chains.size());
    for (size_t c = 0; c < chains.size(); c++) rank_of_chain[c] = rank[chains[c].id];
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
        s->chain_rank[k] = rank_of_chain[br.chain];
        s->model[k] = (uint32_t)chains[br.chain].model_idx;
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
        std::vector<std::pair<uint32_t, uint32_t>> sorted_atoms;
        if (res[r].confs.size() > 1) {  // several conformers: conformer ordinal first, input order inside
            sorted_atoms = res[r].atoms;
            std::stable_sort(sorted_atoms.begin(), sorted_atoms.end(), [](auto &a, auto &b) { return a.first < b.first; });
        }
        const std::vector<std::pair<uint32_t, uint32_t>> &at = res[r].confs.size() > 1 ? sorted_atoms : res[r].atoms;
        ri.atoms.reserve(at.size());
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
    if (s->groups_valid && s->groups_applied == groups) return ARP_OK;  // the attribute words already carry this spec (a pass over every atom otherwise)
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
namespace arp {
// The process-wide default (arp_set_num_threads) is atomic; every table call works on its OWN snapshot (HostThreadsScope), so a
// concurrent arp_set_num_threads / contacts(num_threads=k) on another thread cannot change the slicing between two passes of a call.
static std::atomic<int> g_host_threads{1};
static thread_local int tl_host_threads = 0;  // > 0 inside a HostThreadsScope
static int clamp_threads(int n) {
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(n, 64));
}
int host_threads() { return tl_host_threads > 0 ? tl_host_threads : g_host_threads.load(std::memory_order_relaxed); }
void set_host_threads(int n) { g_host_threads.store(clamp_threads(n), std::memory_order_relaxed); }
HostThreadsScope::HostThreadsScope(int n) : prev(tl_host_threads) {
    tl_host_threads = n < 0 ? g_host_threads.load(std::memory_order_relaxed) : clamp_threads(n);
}
HostThreadsScope::~HostThreadsScope() { tl_host_threads = prev; }
}  // namespace arp
extern "C" void arp_set_num_threads(int32_t n) { arp::set_host_threads(n); }
extern "C" int32_t arp_get_num_threads(void) { return arp::host_threads(); }

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

extern "C" arp_status arp_structure_load(const char *path, int32_t ignore_zero_occupancy, arp_structure **out) try {
    if (!path || !out) { set_error("null argument"); return ARP_ERR_BAD_INPUT; }
    *out = nullptr;
    std::vector<Record> recs;
    std::string p(path), ext;
    size_t dot = p.find_last_of('.');
    if (dot != std::string::npos) ext = p.substr(dot + 1);
    for (char &c : ext) c = (char)tolower((unsigned char)c);
    auto T0 = std::chrono::steady_clock::now();
    arp_status st = (ext == "cif" || ext == "mmcif") ? read_mmcif(path, &recs) : read_pdb(path, &recs);
    if (st != ARP_OK) return st;
    auto T1 = std::chrono::steady_clock::now();
    arp_structure *s = new arp_structure();
    st = build(recs, 0, ignore_zero_occupancy != 0, s);
    auto T2 = std::chrono::steady_clock::now();
    if (g_debug.timing) fprintf(stderr, "read %.3f ms build %.3f ms\n", std::chrono::duration<double, std::milli>(T1 - T0).count(), std::chrono::duration<double, std::milli>(T2 - T1).count());
    if (st != ARP_OK) { delete s; return st; }
    *out = s;
    return ARP_OK;
} ARP_ABI_CATCH

extern "C" arp_status arp_structure_from_records(const arp_records *rec, int32_t hierarchy, arp_structure **out) try {
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
} ARP_ABI_CATCH

extern "C" void arp_structure_free(arp_structure *s) { delete s; }
extern "C" uint64_t arp_structure_n_atoms(const arp_structure *s) { return s ? s->n : 0; }

extern "C" arp_status arp_structure_atoms(arp_structure *s, const char *groups, arp_atoms *v) try {
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
} ARP_ABI_CATCH

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
