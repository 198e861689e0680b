// The library's diagnostic switches (arp_debug_set, include/arpeggia_amd.h).  Shared by the kernel launchers and the host sources.
#pragma once
namespace arp {
// Set through arp_debug_set: the library reads no switch from the environment.
struct DebugKnobs {
    int timing;          // stage laps of the table / batch / ingest paths on stderr
    int emit_kernel;     // 1: the single-pass emitter runs k_pairs<kEmit> (both operands gathered: the route of inputs beyond 2^24 slots) -- parity suite
    long defer_entries;  // > 0: entries of the deferred-probe list of workspaces allocated from now on (tests: a tiny list makes the grow-and-repeat path run)
    int strip_rows;      // > 0: rows per y strip of the cell order (a power of two) for parameter blocks built from now on; 0: chosen by input size (grid.inl grid_setup)
    int table_host;      // test-only library (-DARP_WITH_HOST_TABLE): arp_get_contacts assembles the table on the host
};
extern DebugKnobs g_debug;

}  // namespace arp
