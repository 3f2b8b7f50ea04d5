// vs_node_order_host: a numbering of the graph's nodes that runs along its paths (host only, no device).
//
// node_mat / short_mat (VStrains_PE_Inference.py:139-140) are indexed by the position of a node in the GFA, which an
// assembler assigns as it pleases.  Nothing the path computes depends on that numbering -- counts are sums -- but the
// device's caches do: read pairs are processed in the order of the node their forward read starts in, a slice of that
// order per XCD, so that an L2 holds the postings, node text and matrix cells of one stretch of the graph.  With a
// numbering that scatters neighbours (configs[4], nodes shuffled) k_pe_tiles takes 38.5 instead of 30.4 ms, the counter
// kernel 52 instead of 43, the overflow kernel 11.2 instead of 5.9 (tools/order_probe.py).  So the host side numbers the
// nodes itself, builds the index in that order and maps results back (vstrains_amd/pe.py: Context.build_index,
// PeCounter.result); a C caller can do the same with this function.
//
// The order: depth-first along k-base overlaps (suffix of one oriented node = prefix of the next, either strand -- what
// the assembler's L lines say, derived from the text because PE inference is handed the S lines only,
// PE_Inference.py:100-112).  A node is followed by one of its successors, so numbers run along paths: the first walk
// crosses the whole component, later ones are the branches it left out, each numbered next to where it rejoins.
// vstrains_amd/node_order.py:path_order is the same statement in Python (tests compare the two).
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/vstrains_hip.h"

static inline char comp(char c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return c;
    }
}

extern "C" int vs_node_order_host(const uint8_t *node_ascii, const uint64_t *node_off, uint32_t n_nodes, uint32_t ksize,
                                  uint32_t *order_out) {
    if (!node_off || !order_out || (!node_ascii && n_nodes && node_off[n_nodes])) return VS_E_ARG;
    if (n_nodes > 0x7FFFFFFFu) return VS_E_RANGE;
    const size_t k = ksize;
    std::vector<std::string> fw(n_nodes), rc(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        const size_t len = (size_t)(node_off[i + 1] - node_off[i]);
        fw[i].assign((const char *)node_ascii + node_off[i], len);
        rc[i].resize(len);
        for (size_t p = 0; p < len; p++) rc[i][p] = comp(fw[i][len - 1 - p]);
    }
    // oriented node 2i: the node as given, 2i + 1: its reverse complement; heads[first k bases] = oriented nodes, by number
    std::unordered_map<std::string_view, std::vector<uint32_t>> heads;
    heads.reserve((size_t)n_nodes * 2u + 16u);
    if (k > 0)
        for (uint32_t i = 0; i < n_nodes; i++) {
            if (fw[i].size() < k) continue;
            heads[std::string_view(fw[i]).substr(0, k)].push_back(2u * i);
            heads[std::string_view(rc[i]).substr(0, k)].push_back(2u * i + 1u);
        }
    std::vector<uint8_t> seen(n_nodes, 0);
    std::vector<uint32_t> stack;
    uint32_t n_out = 0;
    for (uint32_t start = 0; start < n_nodes; start++) {
        if (seen[start]) continue;
        stack.push_back(2u * start);
        while (!stack.empty()) {
            const uint32_t t = stack.back();
            stack.pop_back();
            const uint32_t i = t >> 1;
            if (seen[i]) continue;
            seen[i] = 1;
            order_out[n_out++] = i;
            const std::string &f = (t & 1u) ? rc[i] : fw[i], &b = (t & 1u) ? fw[i] : rc[i];
            if (k == 0 || f.size() < k) continue;
            // what precedes this node goes under what follows it: the walk continues forwards first
            auto hit = heads.find(std::string_view(b).substr(b.size() - k, k));
            if (hit != heads.end())
                for (size_t q = hit->second.size(); q-- > 0;)
                    if (!seen[hit->second[q] >> 1]) stack.push_back(hit->second[q] ^ 1u);
            hit = heads.find(std::string_view(f).substr(f.size() - k, k));
            if (hit != heads.end())
                for (size_t q = hit->second.size(); q-- > 0;)
                    if (!seen[hit->second[q] >> 1]) stack.push_back(hit->second[q]);
        }
    }
    return VS_OK;
}
