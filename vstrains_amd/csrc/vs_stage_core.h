// Containers and small numeric helpers of the native stage graph (vs_stage.cpp).  Host-only C++17.
//
// The stages are compared with the reference byte for byte, and the reference's results depend on the behaviour of
// the Python objects it is written with: dicts iterate in insertion order and a popped key that is set again goes to
// the back; sets of small integers iterate in hash-table order; floats print with the shortest digits that round
// trip; numpy sums pairwise.  The types here restate exactly those behaviours, nothing more.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <string>
#include <unordered_map>
#include <vector>

namespace vsg {

typedef uint32_t Nid;  // interned name (node id, contig name, strain name)
static const Nid NO_NID = 0xFFFFFFFFu;

struct StageError {
    int code;
    std::string kind;  // the Python exception the same situation raises in the reference: KeyError, FloatingPointError, ...
    std::string msg;
};

// ---- interned names ------------------------------------------------------------------------------------------------
struct Names {
    std::vector<std::string> str;
    std::unordered_map<std::string, Nid> idx;
    size_t bytes = 0;  // characters of all names (each is held twice: here and as the map's key)
    Nid intern(const std::string &s) {
        auto it = idx.find(s);
        if (it != idx.end()) return it->second;
        Nid n = (Nid)str.size();
        str.push_back(s);
        idx.emplace(s, n);
        bytes += s.size();
        return n;
    }
    Nid find(const std::string &s) const {
        auto it = idx.find(s);
        return it == idx.end() ? NO_NID : it->second;
    }
    const std::string &operator[](Nid n) const { return str[n]; }
    size_t size() const { return str.size(); }
};

// ---- dict keyed by a name: insertion order, pop + set = move to the back -----------------------------------------------
template <class V>
struct NameMap {
    struct Ent {
        Nid k;
        V v;
        bool live;
    };
    std::vector<Ent> ents;
    std::vector<int32_t> slot;  // by name: index into ents, -1 = absent
    size_t n_live = 0;

    size_t size() const { return n_live; }
    bool has(Nid k) const { return k < slot.size() && slot[k] >= 0; }
    V *get(Nid k) { return has(k) ? &ents[slot[k]].v : nullptr; }
    const V *get(Nid k) const { return has(k) ? &ents[slot[k]].v : nullptr; }
    V &set(Nid k, V v) {
        if (k >= slot.size()) slot.resize((size_t)k + 1 + slot.size() / 2, -1);
        if (slot[k] >= 0) {
            ents[slot[k]].v = std::move(v);
            return ents[slot[k]].v;
        }
        slot[k] = (int32_t)ents.size();
        ents.push_back(Ent{k, std::move(v), true});
        n_live++;
        return ents.back().v;
    }
    bool pop(Nid k, V *out = nullptr) {
        if (!has(k)) return false;
        Ent &e = ents[slot[k]];
        if (out) *out = std::move(e.v);
        e.v = V();
        e.live = false;
        slot[k] = -1;
        n_live--;
        return true;
    }
    void clear() {
        for (const Ent &e : ents)
            if (e.live) slot[e.k] = -1;
        ents.clear();
        n_live = 0;
    }
    // drop the dead entries (never while an index-based iteration is running)
    void compact() {
        if (ents.size() == n_live) return;
        size_t w = 0;
        for (size_t r = 0; r < ents.size(); r++) {
            if (!ents[r].live) continue;
            if (w != r) ents[w] = std::move(ents[r]);
            slot[ents[w].k] = (int32_t)w;
            w++;
        }
        ents.resize(w);
    }
    std::vector<Nid> keys() const {
        std::vector<Nid> out;
        out.reserve(n_live);
        for (const Ent &e : ents)
            if (e.live) out.push_back(e.k);
        return out;
    }
};

// ---- open-address index: 64-bit key -> 32-bit value ---------------------------------------------------------------------
struct FlatIdx {
    static constexpr uint64_t EMPTY = 0xFFFFFFFFFFFFFFFFull, TOMB = 0xFFFFFFFFFFFFFFFEull;
    std::vector<uint64_t> keys;
    std::vector<uint32_t> vals;
    size_t used = 0, filled = 0;  // live keys; live + tombstones
    static uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void reserve_for(size_t n) {
        size_t cap = 16;
        while (cap < 2 * n + 2) cap <<= 1;
        if (cap > keys.size()) rehash(cap);
    }
    void rehash(size_t cap) {
        std::vector<uint64_t> ok;
        std::vector<uint32_t> ov;
        ok.swap(keys);
        ov.swap(vals);
        keys.assign(cap, EMPTY);
        vals.assign(cap, 0u);
        used = filled = 0;
        for (size_t i = 0; i < ok.size(); i++)
            if (ok[i] != EMPTY && ok[i] != TOMB) put(ok[i], ov[i]);
    }
    void clear() {
        std::fill(keys.begin(), keys.end(), EMPTY);
        used = filled = 0;
    }
    int64_t find(uint64_t k) const {
        if (keys.empty()) return -1;
        const size_t mask = keys.size() - 1;
        for (size_t i = (size_t)mix(k) & mask;; i = (i + 1) & mask) {
            if (keys[i] == k) return (int64_t)i;
            if (keys[i] == EMPTY) return -1;
        }
    }
    bool get(uint64_t k, uint32_t *v) const {
        int64_t i = find(k);
        if (i < 0) return false;
        *v = vals[i];
        return true;
    }
    void put(uint64_t k, uint32_t v) {
        if (keys.empty() || 2 * (filled + 1) > keys.size()) rehash(keys.empty() ? 16 : (2 * (used + 1) > keys.size() / 2 ? keys.size() * 2 : keys.size()));
        const size_t mask = keys.size() - 1;
        size_t tomb = (size_t)-1;
        for (size_t i = (size_t)mix(k) & mask;; i = (i + 1) & mask) {
            if (keys[i] == k) {
                vals[i] = v;
                return;
            }
            if (keys[i] == TOMB && tomb == (size_t)-1) tomb = i;
            if (keys[i] == EMPTY) {
                if (tomb != (size_t)-1) i = tomb; else filled++;
                keys[i] = k;
                vals[i] = v;
                used++;
                return;
            }
        }
    }
    bool erase(uint64_t k) {
        int64_t i = find(k);
        if (i < 0) return false;
        keys[i] = TOMB;
        used--;
        return true;
    }
};

inline uint64_t pair_key(Nid a, Nid b) { return ((uint64_t)a << 32) | b; }
inline Nid key_first(uint64_t k) { return (Nid)(k >> 32); }
inline Nid key_second(uint64_t k) { return (Nid)k; }

// ---- dict keyed by a pair of names --------------------------------------------------------------------------------------
// Entries in insertion order plus an open-address table of 32-bit entry numbers (0 = empty, ~0 = deleted): a lookup
// costs one probe into a table a quarter the size of a key table, and the keys are read where the entries are.  The
// table is built by the first lookup after `clear` / `append_new`: a map that is filled and only iterated (the edge map
// of a re-initialised stage graph between two extracted paths) never pays for it.
template <class V>
struct PairMap {
    struct Ent {
        uint64_t k;
        V v;
        bool live;
    };
    std::vector<Ent> ents;
    mutable std::vector<uint32_t> tab;
    mutable size_t n_filled = 0;   // occupied slots (live + deleted)
    mutable bool tab_valid = true;  // the table holds every live entry
    size_t n_live = 0;
    static const uint32_t DELETED = 0xFFFFFFFFu;

    size_t size() const { return n_live; }
    void rebuild_table(size_t cap) const {
        tab.assign(cap, 0u);
        n_filled = 0;
        const size_t mask = cap - 1;
        for (size_t e = 0; e < ents.size(); e++) {
            if (!ents[e].live) continue;
            size_t i = (size_t)FlatIdx::mix(ents[e].k) & mask;
            while (tab[i]) i = (i + 1) & mask;
            tab[i] = (uint32_t)e + 1u;
            n_filled++;
        }
        tab_valid = true;
    }
    void need_table() const {
        if (tab_valid) return;
        size_t cap = tab.size() < 16 ? 16 : tab.size();
        while (cap < 4 * (n_live + 1)) cap <<= 1;
        rebuild_table(cap);
    }
    // slot of key k, or -1
    int64_t find_slot(uint64_t k) const {
        need_table();
        if (tab.empty()) return -1;
        const size_t mask = tab.size() - 1;
        for (size_t i = (size_t)FlatIdx::mix(k) & mask;; i = (i + 1) & mask) {
            const uint32_t t = tab[i];
            if (t == 0u) return -1;
            if (t != DELETED && ents[t - 1u].k == k) return (int64_t)i;
        }
    }
    bool has(uint64_t k) const { return find_slot(k) >= 0; }
    V *get(uint64_t k) {
        const int64_t i = find_slot(k);
        return i >= 0 ? &ents[tab[i] - 1u].v : nullptr;
    }
    const V *get(uint64_t k) const {
        const int64_t i = find_slot(k);
        return i >= 0 ? &ents[tab[i] - 1u].v : nullptr;
    }
    void set(uint64_t k, V v) {
        need_table();
        if (tab.empty() || 2 * (n_filled + 1) > tab.size()) {
            size_t cap = tab.empty() ? 16 : tab.size();
            while (cap < 4 * (n_live + 1)) cap <<= 1;
            rebuild_table(cap);
        }
        const size_t mask = tab.size() - 1;
        size_t at = (size_t)-1;
        for (size_t i = (size_t)FlatIdx::mix(k) & mask;; i = (i + 1) & mask) {
            const uint32_t t = tab[i];
            if (t == 0u) {
                if (at == (size_t)-1) { at = i; n_filled++; }
                break;
            }
            if (t == DELETED) {
                if (at == (size_t)-1) at = i;
                continue;
            }
            if (ents[t - 1u].k == k) {
                ents[t - 1u].v = std::move(v);
                return;
            }
        }
        tab[at] = (uint32_t)ents.size() + 1u;
        ents.push_back(Ent{k, std::move(v), true});
        n_live++;
    }
    // a key the caller knows to be absent (the edges of a re-initialised graph, each once): no lookup, the table is left
    // to the first reader
    void append_new(uint64_t k, V v) {
        ents.push_back(Ent{k, std::move(v), true});
        n_live++;
        tab_valid = false;
    }
    bool pop(uint64_t k, V *out = nullptr) {
        const int64_t i = find_slot(k);
        if (i < 0) return false;
        Ent &e = ents[tab[i] - 1u];
        if (out) *out = std::move(e.v);
        e.live = false;
        tab[i] = DELETED;
        n_live--;
        return true;
    }
    void clear() {
        ents.clear();
        n_live = 0;
        n_filled = 0;
        tab_valid = false;  // (the stale table is overwritten by the rebuild the first lookup asks for)
    }
    void reserve(size_t n) { ents.reserve(n); }
    void compact() {
        if (ents.size() == n_live) return;
        size_t w = 0;
        for (size_t r = 0; r < ents.size(); r++) {
            if (!ents[r].live) continue;
            if (w != r) ents[w] = std::move(ents[r]);
            w++;
        }
        ents.resize(w);
        tab_valid = false;
    }
    std::vector<uint64_t> keys() const {
        std::vector<uint64_t> out;
        out.reserve(n_live);
        for (const Ent &e : ents)
            if (e.live) out.push_back(e.k);
        return out;
    }
};

// ---- CPython's set of small non-negative ints: iteration order -----------------------------------------------------------
// The reference iterates `set(vertex.in_neighbors())` (Decomposition.py:561,625); hash(Vertex) = its index, and a set
// iterates its hash table slot by slot.  Objects/setobject.c (3.8 - 3.12): table of 8 slots, linear probing over up to
// 9 following slots, then i = i * 5 + 1 + (perturb >>= 5); the table grows to the first power of two above 4 * used as
// soon as fill * 5 >= mask * 3, re-inserting the old table in slot order.
struct PyIntSet {
    std::vector<int64_t> table;  // -1 = unused
    size_t mask = 7, fill = 0;
    PyIntSet() : table(8, -1) {}
    static void insert_clean(std::vector<int64_t> &t, size_t mask, int64_t key) {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            if (t[i] < 0) { t[i] = key; return; }
            if (i + 9 <= mask) {
                for (size_t j = 1; j <= 9; j++)
                    if (t[i + j] < 0) { t[i + j] = key; return; }
            }
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void add(int64_t key) {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            size_t probes = (i + 9 <= mask) ? 9 : 0;
            size_t e = i;
            bool placed = false;
            for (;;) {
                if (table[e] < 0) { table[e] = key; placed = true; break; }
                if (table[e] == key) return;
                if (!probes) break;
                probes--;
                e++;
            }
            if (placed) break;
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
        fill++;
        if (fill * 5 < mask * 3) return;
        size_t minused = fill > 50000 ? fill * 2 : fill * 4, newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<int64_t> nt(newsize, -1);
        for (int64_t k : table)
            if (k >= 0) insert_clean(nt, newsize - 1, k);
        table.swap(nt);
        mask = newsize - 1;
    }
    std::vector<uint32_t> order() const {
        std::vector<uint32_t> out;
        for (int64_t k : table)
            if (k >= 0) out.push_back((uint32_t)k);
        return out;
    }
};

// ---- float text ---------------------------------------------------------------------------------------------------------
// repr(float) / str(float) of CPython (format code 'r'): the shortest digits that round trip; exponent form when the
// decimal exponent is < -4 or >= 16; ".0" appended to integral values in positional form.
inline std::string py_repr(double x) {
    if (std::isnan(x)) return "nan";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);
    std::string s(buf, r.ptr);  // [-]d[.ddd]e[+-]XX
    std::string out;
    size_t p = 0;
    if (s[0] == '-') { out.push_back('-'); p = 1; }
    size_t epos = s.find('e', p);
    std::string mant = s.substr(p, epos - p);
    int exp10 = atoi(s.c_str() + epos + 1);
    std::string digits;
    for (char c : mant)
        if (c != '.') digits.push_back(c);
    const int nd = (int)digits.size();
    const int decpt = exp10 + 1;  // position of the decimal point relative to the digits
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            out += "0.";
            out.append((size_t)(-decpt), '0');
            out += digits;
        } else if (decpt >= nd) {
            out += digits;
            out.append((size_t)(decpt - nd), '0');
            out += ".0";
        } else {
            out.append(digits, 0, (size_t)decpt);
            out.push_back('.');
            out.append(digits, (size_t)decpt, std::string::npos);
        }
    } else {
        out.push_back(digits[0]);
        if (nd > 1) {
            out.push_back('.');
            out.append(digits, 1, std::string::npos);
        }
        out.push_back('e');
        int e = decpt - 1;
        out.push_back(e < 0 ? '-' : '+');
        if (e < 0) e = -e;
        char eb[16];
        snprintf(eb, sizeof(eb), "%02d", e);
        out += eb;
    }
    return out;
}

// round(x, 2) of a Python float: the correctly rounded two-decimal string of the exact binary value, read back
inline double py_round2(double x) {
    if (!std::isfinite(x)) return x;
    char buf[512];
    snprintf(buf, sizeof(buf), "%.2f", x);
    return strtod(buf, nullptr);
}

// ---- numpy sums ---------------------------------------------------------------------------------------------------------
// numpy.sum / numpy.mean of a contiguous float64 vector: pairwise_sum of numpy/_core/src/umath/loops_utils.h.src on top
// of the reduction's identity 0.0.
inline double np_pairwise(const double *a, size_t n) {
    if (n < 8) {
        double res = 0.;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        size_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
inline double np_sum(const std::vector<double> &a) { return 0.0 + np_pairwise(a.data(), a.size()); }
inline double np_mean(const std::vector<double> &a) { return np_sum(a) / (double)a.size(); }
// numpy.median of a non-empty list without NaNs: mean of the one or two middle elements of the sorted values
inline double np_median(std::vector<double> a) {
    const size_t n = a.size();
    if (n == 0) return std::nan("");
    std::sort(a.begin(), a.end());
    if (n & 1) return (0.0 + a[n / 2]) / 1.0;
    return ((0.0 + a[n / 2 - 1]) + a[n / 2]) / 2.0;
}

}  // namespace vsg
