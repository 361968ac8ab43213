// Order of `list(set(idle) & set(machines))` for machine indices < 32, as
// CPython 3.10 produces it (environments/SO_FJSSP.py:302-303 builds its
// candidate lists that way, and max()/min()/random.choice then depend on the
// order).  Closed form of Objects/setobject.c for this value range:
//
//   * a set of small ints that ever held >= 5 elements was resized to a table of
//     >= 32 slots where every value < 32 sits in its own slot: ascending order;
//   * a set of <= 4 elements lives in the initial 8-slot table: slot = v & 7,
//     collisions re-probe with i = (5 i + 1 + (perturb >>= 5)) & 7 (no linear
//     probes in an 8-slot table), so the order depends on insertion order --
//     unless every value is < 8, where slot = value and the order is ascending;
//   * set & set iterates the SMALLER operand (the right one on a tie) in its
//     table order, keeps members of the other, and inserts them into a fresh set.
//
// Compiled for both host and device: the kernels use it in machine_select, and
// tests call the host build through fjsp_pyset_and_order() to compare it with
// the running interpreter.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FJSP_HD __host__ __device__
#else
#define FJSP_HD
#endif

namespace fjsp {

struct CandList {
    uint32_t mask;    // the candidates as a bit set
    uint32_t packed;  // when !asc: up to 4 machine ids, 8 bits each, in iteration order
    int n;
    bool asc;         // iterate the mask in ascending bit order
};

FJSP_HD inline int popc32(uint32_t v) { return __builtin_popcount(v); }

// insert v into an 8-slot table kept as 8 bytes (0xFF = empty)
FJSP_HD inline uint64_t pyset8_insert(uint64_t table, uint32_t v) {
    uint32_t i = v & 7u, perturb = v;
    for (;;) {
        if (((table >> (8 * i)) & 0xFFull) == 0xFFull)
            return (table & ~(0xFFull << (8 * i))) | ((uint64_t)v << (8 * i));
        perturb >>= 5;
        i = (i * 5u + 1u + perturb) & 7u;
    }
}

// i-th candidate in iteration order
FJSP_HD inline int cand_at(const CandList &c, int i) {
    if (!c.asc) return (int)((c.packed >> (8 * i)) & 0xFFu);
    uint32_t m = c.mask;
    for (int q = 0; q < i; ++q) m &= m - 1;
    return __builtin_ctz(m);
}

// A = set(machine_idle_list) (inserted in ascending order); B = set(machine tuple):
// inserted in ascending order when b_ascending, else in the order of b_first4
// (only consulted when B has <= 4 elements).
FJSP_HD inline CandList pyset_and(uint32_t a_mask, uint32_t b_mask, uint32_t b_first4, bool b_ascending) {
    CandList out;
    out.mask = a_mask & b_mask;
    out.n = popc32(out.mask);
    out.packed = 0;
    out.asc = true;
    if (out.n >= 5 || out.n <= 1) return out;
    // every member of both operands < 8: in an 8-slot table value v sits in slot v (no collision is possible
    // between distinct values < 8), so every table involved iterates in ascending order
    if (((a_mask | b_mask) >> 8) == 0) return out;
    const int nA = popc32(a_mask), nB = popc32(b_mask);
    const bool iter_a = nB > nA;                       // set_intersection swaps to iterate the smaller operand
    const uint32_t it_mask = iter_a ? a_mask : b_mask;
    const int it_n = iter_a ? nA : nB;
    // filtered insertion sequence of the result set (<= 4 entries)
    uint32_t seq = 0;
    int ns = 0;
    if (it_n >= 5) {                                   // iterated set is in ascending order
        uint32_t m = out.mask;
        while (m) { seq |= (uint32_t)__builtin_ctz(m) << (8 * ns++); m &= m - 1; }
    } else {
        uint64_t table = ~0ull;
        if (iter_a || b_ascending) {
            uint32_t m = it_mask;
            while (m) { table = pyset8_insert(table, (uint32_t)__builtin_ctz(m)); m &= m - 1; }
        } else {
            for (int q = 0; q < it_n; ++q) table = pyset8_insert(table, (b_first4 >> (8 * q)) & 0xFFu);
        }
        for (int sl = 0; sl < 8; ++sl) {
            const uint32_t v = (uint32_t)((table >> (8 * sl)) & 0xFFull);
            if (v != 0xFFu && ((out.mask >> v) & 1u)) seq |= v << (8 * ns++);
        }
    }
    uint64_t table = ~0ull;
    for (int q = 0; q < ns; ++q) table = pyset8_insert(table, (seq >> (8 * q)) & 0xFFu);
    int no = 0;
    for (int sl = 0; sl < 8; ++sl) {
        const uint32_t v = (uint32_t)((table >> (8 * sl)) & 0xFFull);
        if (v != 0xFFu) out.packed |= v << (8 * no++);
    }
    out.asc = false;
    return out;
}

}  // namespace fjsp
