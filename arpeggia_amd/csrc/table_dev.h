// Device side of the contact table (SURVEY.md 8f rows f1 + f2): plane fits, ring rows, row expansion, the 10-key sort and the
// side-chain plane statistics run as HIP kernels (table_dev.hip); table.cpp keeps the bookkeeping (entity lists, strings).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/arpeggia_amd.h"

namespace arp {

// One ring of the output vocabulary (complex.rs:334-342: entity "Ring", atomi 0), as the device needs it.
struct RingEnt {
    uint32_t src_res;        // residue whose ring atoms define the plane
    int32_t model_serial;    // the model serial the ring is filed under (complex.rs:447-449)
    uint32_t model_rank;     // rank of that serial among the file's serials (sort key `model`)
    uint32_t chain_rank;
    uint32_t flags;          // 1 = chain in the ligand set, 2 = in the receptor set, 4 = resolves to a residue (has an ordinal)
    uint32_t ord;            // positional index of that residue in its chain (complex.rs:411-440)
    uint32_t sc_src;         // residue whose side-chain plane applies to the ring entity, or ARP_NONE
    uint32_t pad;
};

struct EntKey { int32_t resi; uint32_t altloc; int32_t atomi; uint32_t icode; };  // names as big-endian words: unsigned compare == byte-wise order

// Device-resident copy of a structure (kept with the arp_structure, uploaded once): the arp_atoms arrays + what the table kernels need.
struct DevStructure {
    int device = -1;
    char *block = nullptr;               // one allocation
    uint64_t n = 0, n_res = 0, n_h = 0;
    double *x = nullptr, *y = nullptr, *z = nullptr;
    uint32_t *attr = nullptr, *res_ord = nullptr, *res_id = nullptr, *res_h_ptr = nullptr, *res_h_idx = nullptr, *res_cb = nullptr, *res_sg = nullptr;
    uint32_t *chain_rank = nullptr, *model = nullptr;
    uint8_t *plane_bits = nullptr;       // per atom: 1 = ring-plane atom (residues.rs:163-186), 2 = sc-plane atom (residues.rs:188-268)
    uint32_t *res_atom_ptr = nullptr, *res_atom_idx = nullptr;   // residue -> atoms in hierarchy order
    uint32_t *atom_sc_src = nullptr;     // per atom: residue whose sc plane applies, or ARP_NONE
    EntKey *ent_key = nullptr;           // per atom
    uint32_t *model_rank = nullptr;      // per model ordinal
    int32_t *model_serial_of = nullptr;  // per model ordinal
    // derived once per structure by the first device_table call (they depend on the structure alone): the fitted planes and the entity ranks
    char *derived = nullptr;             // one allocation
    void *ring_pl = nullptr, *sc_pl = nullptr; uint8_t *pl_valid = nullptr; uint32_t *ent_rank = nullptr; uint64_t derived_n_ent = 0;
    uint32_t max_ent_rank = 0;           // largest rank of an entity inside its chain (the width of a rank in the rows' sort key)
    uint32_t n_chains = 0, n_models = 0;  // distinct chain ids / models of the structure (widths of the sort keys)
    bool any_icode = false;              // some atom carries an insertion code (otherwise that sort pass is skipped)
    std::string attr_groups;             // the chain groups the resident attr words were built for
    char *rings_block = nullptr;         // the ring entities {RingEnt[], EntKey[]} on the device; sent again when they differ from rings_host (table_dev.hip)
    uint64_t rings_cap = 0;
    std::vector<char> rings_host;
};

struct TableRow { uint32_t from_ent, to_ent; float distance; int32_t interaction; };   // entity = atom index, or n_atoms + ring index
struct TableSc { float dist, dihedral, angle, valid; };                                   // valid != 0: both residues have a side-chain plane
// What comes back: rows in final order.  A large table lands straight in a pooled pinned block (engine.cpp pinned_block) that the table
// then OWNS -- no copy out of a landing buffer; a small one is copied into plain arrays.  `owner` keeps whichever it is alive.
struct TableRowsHost {
    uint64_t n = 0;
    std::shared_ptr<char> owner;
    TableRow *rows = nullptr;
    TableSc *sc = nullptr;
};
std::shared_ptr<char> pinned_block(size_t bytes);   // pooled pinned host memory; the deleter gives the block back to the pool (engine.cpp)

// Runs the whole device pipeline for one structure.  `pairs_dev` = contacts-only pair list on the device (arp_contacts_atomic,
// ARP_MEM_DEVICE).  Returns ARP_ERR_NO_RINGS etc. like arp_get_contacts.
arp_status device_table(arp_context *ctx, DevStructure &ds, const std::vector<RingEnt> &rings, const std::vector<EntKey> &ring_keys,
                        const arp_pair *pairs_dev, uint64_t n_pairs, double dist_cutoff, TableRowsHost *out);
// the planes the device fitted, for tests (PHE4 of 1ubq: residues.rs:355-372): 12 doubles per residue {ring c, ring n, sc c, sc n} + validity bits
arp_status device_planes(arp_context *ctx, const DevStructure &ds, std::vector<double> *planes, std::vector<uint8_t> *valid);

// engine.cpp: the context's stream / device and grow-only scratch (device and pinned host)
void *context_stream(arp_context *ctx);
int context_device(arp_context *ctx);
// engine.cpp: the table path's pair pass -- device-resident inputs, *data = a view of the context's own pair buffer (valid until its next call)
arp_status contacts_atomic_view(arp_context *ctx, const arp_atoms *atoms, const arp_params *params, const arp_pair **data, uint64_t *n);
struct GridParams;
struct Fat;
// the cell list the most recent pair pass of this context built -- valid only if that pass ran on exactly these arrays (x, n)
bool context_grid(arp_context *ctx, const double *x, uint64_t n, const GridParams **grid, const uint32_t **cell_start, const Fat **fat);
arp_status context_scratch(arp_context *ctx, int slot, uint64_t dev_bytes, uint64_t pinned_bytes, char **dev, char **pinned);  // slot 0 / 1: two independent grow-only blocks

}  // namespace arp
