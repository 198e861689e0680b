"""Round-4 ablation builds of k_emit (results wrong by construction; bench.py --no-check): what each wait of the exact phase costs.
Usage: python tests/microbench/ablate_r4.py   (builds tests/microbench/build/libvar_<name>.so through build_variant.py)"""
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
VARIANTS = {
    # neighbour gathers from the first 4 KB of the record array (always in the L1): no L2 / HBM round trip
    "nogather": ['pairs_emit.inl::    const char *gp = reinterpret_cast<const char *>(so.fat) + (size_t)goff;::    const char *gp = reinterpret_cast<const char *>(so.fat) + (size_t)(goff & 0xFF0u);'],
    # only ONE of the three gather loads (x, y of the neighbour); z, pair word, index and key faked from the entry: cuts the vector-memory
    # address work by two thirds while nearly every candidate stays valid (dz = 0 only shortens the distance)
    "gather1": ['pairs_emit.inl::    g.bxy = *reinterpret_cast<const u32x4 *>(gp); g.bzp = *reinterpret_cast<const u32x4 *>(gp + 16);\n    g.kb = *reinterpret_cast<const unsigned long long *>(gp + 32);::    g.bxy = *reinterpret_cast<const u32x4 *>(gp); g.bzp = u32x4{0u, 0u, 0x00051505u, g.e & 0xFFFFFFu};\n    g.kb = ((unsigned long long)(g.e >> 24) << 40) | (g.e & 0xFFFFFFu);',
                'pairs_emit.inl::    const u32x4 bxy = g.bxy, bzp = g.bzp;::    const u32x4 bxy = g.bxy; u32x4 bzp = g.bzp;',
                'pairs_emit.inl::    const uint32_t pa = azp.z, pb = bzp.z;::    bzp.x = azp.x; bzp.y = azp.y;\n    const uint32_t pa = azp.z, pb = bzp.z;'],
    # two of the three (x, y and z, pair word, index); the key faked
    "gather2": ['pairs_emit.inl::    g.kb = *reinterpret_cast<const unsigned long long *>(gp + 32);::    g.kb = ((unsigned long long)(g.e >> 24) << 40) | (g.e & 0xFFFFFFu);'],
    # (not an ablation: results stay right)  the exact record padded to one 64-byte line per atom: a gather touches 1 line instead of ~1.5
    "fat64": ['arp_internal.h::struct __attribute__((aligned(16))) Fat {::struct __attribute__((aligned(64))) Fat {',
              'pairs.inl::    asm("v_lshl_add_u32 %0, %1, 1, %1\\n\\tv_lshlrev_b32 %0, 4, %0" : "=v"(off) : "v"(p));::    asm("v_lshlrev_b32 %0, 6, %1" : "=v"(off) : "v"(p));',
              'pairs_emit.inl::    asm("v_mul_u32_u24 %0, %1, 48" : "=v"(goff) : "v"(g.e));::    asm("v_mul_u32_u24 %0, %1, 64" : "=v"(goff) : "v"(g.e));'],
    # (results stay right)  the gathers as non-temporal loads (L1 bypass)
    "gnt": ['pairs_emit.inl::    g.bxy = *reinterpret_cast<const u32x4 *>(gp); g.bzp = *reinterpret_cast<const u32x4 *>(gp + 16);\n    g.kb = *reinterpret_cast<const unsigned long long *>(gp + 32);::    g.bxy = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(gp)); g.bzp = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(gp + 16));\n    g.kb = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(gp + 32));'],
    # (results stay right)  every gather issued twice (the copies land in registers nobody reads): is the vector-memory address path the limit?
    "gather2x": ['pairs_emit.inl::    g.kb = *reinterpret_cast<const unsigned long long *>(gp + 32);\n    return g;::    g.kb = *reinterpret_cast<const unsigned long long *>(gp + 32);\n    { u32x4 d0, d1; u32x2 d2; asm volatile("global_load_dwordx4 %0, %3, %4\\n\\tglobal_load_dwordx4 %1, %3, %4 offset:16\\n\\tglobal_load_dwordx2 %2, %3, %4 offset:32\\n\\ts_waitcnt vmcnt(0)" : "=&v"(d0), "=&v"(d1), "=&v"(d2) : "v"(goff), "s"(so.fat) : "memory"); }\n    return g;'],
    # (results stay right)  every prefilter LDS read issued twice: is the LDS pipe the limit?
    "lds2x": ['pairs_emit.inl::                            float rx[kReadAhead], ry[kReadAhead], rz[kReadAhead], rw[kReadAhead];::                            f32x4 dd; { const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float4 *)(win + g * kEGroup + u0); asm volatile("ds_read_b128 %0, %1\\n\\tds_read_b128 %0, %1 offset:16\\n\\tds_read_b128 %0, %1 offset:32\\n\\tds_read_b128 %0, %1 offset:48" : "=&v"(dd) : "v"(la)); }\n                            float rx[kReadAhead], ry[kReadAhead], rz[kReadAhead], rw[kReadAhead];',
              'pairs_emit.inl::                            for (uint32_t u = 0; u < kReadAhead; ++u) push_sign_e(mask, acc[u], thr);::                            for (uint32_t u = 0; u < kReadAhead; ++u) push_sign_e(mask, acc[u], thr);\n                            asm volatile("" : : "v"(dd));'],
    # (results stay right)  the prefilter arithmetic of every test done twice (second copy into a dead mask): is vector issue the limit?
    "valu2x": ['pairs_emit.inl::                            for (uint32_t u = 0; u < kReadAhead; ++u) push_sign_e(mask, acc[u], thr);::                            for (uint32_t u = 0; u < kReadAhead; ++u) push_sign_e(mask, acc[u], thr);\n                            { uint32_t m2 = mask; float a2[kReadAhead];\n                              for (uint32_t u = 0; u < kReadAhead; ++u) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a2[u]) : "v"(rx[u]), "v"(hm2.x), "v"(rw[u]));\n                              for (uint32_t u = 0; u < kReadAhead; ++u) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2[u]) : "v"(ry[u]), "v"(hm2.y));\n                              for (uint32_t u = 0; u < kReadAhead; ++u) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2[u]) : "v"(rz[u]), "v"(hm2.z));\n                              for (uint32_t u = 0; u < kReadAhead; ++u) push_sign_e(m2, a2[u], thr);\n                              asm volatile("" : : "v"(m2)); }'],
    # no record store
    "nostore": ['pairs_emit.inl::            "global_store_dwordx4 %[off], %[rec], %[base] nt\\n\\t"::            "s_nop 0\\n\\t"'],
    # no rule-table read
    "nolut": ['pairs_emit.inl::    const uint32_t t = tb.lut[(L << 7) | w1 | w2];::    const uint32_t t = ((L << 7) | w1 | w2) & 0x7FFFFu;'],
    # no chunk staging loads
    "nostage": ['pairs_emit.inl::                for (uint32_t p = cs + lane; p < ce; p += 64u) w.nrec[p - cs] = so.rec[p];::                for (uint32_t p = cs + lane; p < ce; p += 64u) w.nrec[p - cs] = make_float4((float)p, 1.f, 2.f, 3.f);'],
}
for name, patches in VARIANTS.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    subprocess.run([sys.executable, str(HERE / "build_variant.py"), name, *[q.replace("\\n    ", "\n    ") for q in patches]], check=True)
