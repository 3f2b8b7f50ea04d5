// FASTQ ingest (host side, multi-threaded): the reference reads both files with readlines() in
// text mode, takes record r as lines 4r..4r+3 and uses line 4r+1 minus its LAST CHARACTER as the
// sequence (utils/VStrains_PE_Inference.py:146-159).  Text mode means universal newlines: "\r\n"
// and a lone "\r" both end a line.  A final line without a newline still loses its last (real)
// character.  total = min(len_f // 4, len_r // 4).
//
// vs_fastq_open maps both files and indexes the sequence lines (newline counting and line
// location split over the host cores); vs_fastq_block gathers the sequences of a record range
// into one interleaved ASCII buffer (forward, reverse, forward, ...) on all cores and hands it to
// vs_reads_pack, which packs on the device.  No Python objects, no per-record allocation.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "vs_internal.h"

namespace {

struct FqFile {
    const uint8_t *data = nullptr;
    size_t size = 0;
    int fd = -1;
    std::vector<uint8_t> translated;   // only when the file holds '\r': universal-newline copy
    std::vector<uint64_t> seq_start;   // per record: first byte of line 4r+1
    std::vector<uint32_t> seq_len;     // ... its length after dropping the last character
    uint64_t n_lines = 0;
    const uint8_t *text() const { return translated.empty() ? data : translated.data(); }
    size_t text_size() const { return translated.empty() ? size : translated.size(); }
};

unsigned n_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char *ev = getenv("VS_HOST_THREADS")) n = (unsigned)atoi(ev);
    return std::max(1u, std::min(n, 64u));
}

template <typename F>
void parallel_for(unsigned parts, F fn) {
    std::vector<std::thread> th;
    for (unsigned p = 1; p < parts; p++) th.emplace_back(fn, p);
    fn(0u);
    for (auto &t : th) t.join();
}

int map_file(vs_ctx *ctx, const char *path, FqFile &f) {
    f.fd = open(path, O_RDONLY);
    if (f.fd < 0) return vs_fail(ctx, VS_E_ARG, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(f.fd, &st) != 0) return vs_fail(ctx, VS_E_ARG, "cannot stat %s: %s", path, strerror(errno));
    f.size = (size_t)st.st_size;
    if (f.size) {
        void *p = mmap(nullptr, f.size, PROT_READ, MAP_PRIVATE, f.fd, 0);
        if (p == MAP_FAILED) return vs_fail(ctx, VS_E_OOM, "cannot map %s: %s", path, strerror(errno));
        f.data = (const uint8_t *)p;
        madvise(p, f.size, MADV_SEQUENTIAL);
    }
    return VS_OK;
}

// Index the sequence lines of one file.
int index_file(vs_ctx *ctx, const char *path, FqFile &f) {
    const unsigned T = n_threads();
    // bytes >= 0x80 would be decoded (or rejected) by Python's text mode; not reproduced
    {
        std::vector<int> bad(T, 0), cr(T, 0);
        parallel_for(T, [&](unsigned p) {
            size_t lo = f.size * p / T, hi = f.size * (p + 1) / T;
            for (size_t i = lo; i < hi; i++) {
                uint8_t c = f.data[i];
                if (c >= 0x80) { bad[p] = 1; break; }
                if (c == '\r') cr[p] = 1;
            }
        });
        for (unsigned p = 0; p < T; p++)
            if (bad[p]) return vs_fail(ctx, VS_E_ARG, "%s holds non-ASCII bytes; the reference's text-mode decoding is not reproduced", path);
        bool any_cr = false;
        for (unsigned p = 0; p < T; p++) any_cr |= cr[p] != 0;
        if (any_cr) {  // rare: make the universal-newline view once, then index that
            f.translated.reserve(f.size);
            for (size_t i = 0; i < f.size; i++) {
                uint8_t c = f.data[i];
                if (c == '\r') {
                    f.translated.push_back('\n');
                    if (i + 1 < f.size && f.data[i + 1] == '\n') i++;
                } else {
                    f.translated.push_back(c);
                }
            }
            if (f.translated.empty() && f.size) f.translated.push_back('\n');
        }
    }
    const uint8_t *txt = f.text();
    const size_t n = f.text_size();
    // pass 1: newlines per part
    std::vector<uint64_t> nl(T + 1, 0);
    parallel_for(T, [&](unsigned p) {
        size_t lo = n * p / T, hi = n * (p + 1) / T;
        uint64_t c = 0;
        const uint8_t *q = txt + lo, *end = txt + hi;
        while (q < end) {
            const uint8_t *r = (const uint8_t *)memchr(q, '\n', (size_t)(end - q));
            if (!r) break;
            c++;
            q = r + 1;
        }
        nl[p + 1] = c;
    });
    for (unsigned p = 0; p < T; p++) nl[p + 1] += nl[p];
    const uint64_t n_newlines = nl[T];
    const bool open_tail = n > 0 && txt[n - 1] != '\n';
    f.n_lines = n_newlines + (open_tail ? 1 : 0);
    const uint64_t n_rec = f.n_lines / 4;
    f.seq_start.assign(n_rec, 0);
    f.seq_len.assign(n_rec, 0);
    // pass 2: the newline that ends line L (0-based) is newline number L; a sequence line has
    // L % 4 == 1; it starts right after newline L-1
    parallel_for(T, [&](unsigned p) {
        size_t lo = n * p / T, hi = n * (p + 1) / T;
        uint64_t line = nl[p];  // index of the line that the next newline in this part ends
        // start of that line: after the previous newline (possibly in an earlier part)
        size_t start = 0;
        if (lo > 0) {
            const uint8_t *r = (const uint8_t *)memrchr(txt, '\n', lo);
            start = r ? (size_t)(r - txt) + 1 : 0;
        }
        const uint8_t *q = txt + lo, *end = txt + hi;
        while (q < end) {
            const uint8_t *r = (const uint8_t *)memchr(q, '\n', (size_t)(end - q));
            if (!r) break;
            if ((line & 3u) == 1u && (line >> 2) < n_rec) {
                f.seq_start[line >> 2] = start;
                f.seq_len[line >> 2] = (uint32_t)((size_t)(r - txt) - start);  // the dropped char is the newline
            }
            start = (size_t)(r - txt) + 1;
            line++;
            q = r + 1;
        }
    });
    if (open_tail) {  // last line without newline: it loses a real character
        const uint64_t line = n_newlines;
        if ((line & 3u) == 1u && (line >> 2) < n_rec) {
            const uint8_t *r = n ? (const uint8_t *)memrchr(txt, '\n', n) : nullptr;
            size_t start = r ? (size_t)(r - txt) + 1 : 0;
            f.seq_start[line >> 2] = start;
            f.seq_len[line >> 2] = (uint32_t)(n - start - 1);
        }
    }
    for (uint64_t r = 0; r < n_rec; r++)
        if (f.seq_len[r] > VS_LEN_MASK) return vs_fail(ctx, VS_E_RANGE, "%s: record %llu has a %u-byte sequence line", path, (unsigned long long)r, f.seq_len[r]);
    return VS_OK;
}

void close_file(FqFile &f) {
    if (f.data) munmap((void *)f.data, f.size);
    if (f.fd >= 0) close(f.fd);
    f.data = nullptr;
    f.fd = -1;
}

}  // namespace

struct vs_fastq {
    FqFile f[2];
    uint64_t n_pairs = 0;
    // pinned staging for vs_fastq_block (grow-only): the gather writes here, the upload reads here
    uint8_t *stage = nullptr;
    size_t stage_cap = 0;
    uint64_t *stage_off = nullptr;
    size_t stage_off_cap = 0;
};

extern "C" {

int vs_fastq_open(vs_ctx *ctx, const char *fwd_path, const char *rve_path, vs_fastq **out) {
    if (!fwd_path || !rve_path || !out) return vs_fail(ctx, VS_E_ARG, "vs_fastq_open: bad argument");
    *out = nullptr;
    vs_fastq *fq = new vs_fastq();
    const char *paths[2] = {fwd_path, rve_path};
    for (int i = 0; i < 2; i++) {
        int rc = map_file(ctx, paths[i], fq->f[i]);
        if (rc == VS_OK) rc = index_file(ctx, paths[i], fq->f[i]);
        if (rc != VS_OK) {
            close_file(fq->f[0]);
            close_file(fq->f[1]);
            delete fq;
            return rc;
        }
    }
    fq->n_pairs = std::min(fq->f[0].seq_start.size(), fq->f[1].seq_start.size());  // PE_Inference.py:154
    *out = fq;
    return VS_OK;
}

void vs_fastq_close(vs_fastq *fq) {
    if (!fq) return;
    if (fq->stage) (void)hipHostFree(fq->stage);
    if (fq->stage_off) (void)hipHostFree(fq->stage_off);
    close_file(fq->f[0]);
    close_file(fq->f[1]);
    delete fq;
}

int vs_fastq_info(const vs_fastq *fq, uint64_t info[3]) {
    if (!fq || !info) return VS_E_ARG;
    info[0] = fq->n_pairs;
    info[1] = fq->f[0].n_lines;
    info[2] = fq->f[1].n_lines;
    return VS_OK;
}

// Sequence of record `record` of file `which` (0 forward, 1 reverse) into buf (cap bytes); *len
// receives its length.  Host only (tests, small tools).
int vs_fastq_sequence(const vs_fastq *fq, int which, uint64_t record, uint8_t *buf, uint32_t cap, uint32_t *len) {
    if (!fq || which < 0 || which > 1 || !len) return VS_E_ARG;
    const FqFile &f = fq->f[which];
    if (record >= f.seq_start.size()) return VS_E_RANGE;
    *len = f.seq_len[record];
    if (buf && cap >= *len && *len) memcpy(buf, f.text() + f.seq_start[record], *len);
    return VS_OK;
}

// Host gather only: interleaved ASCII + offsets of pairs [first, first+count) (tests / callers
// that want the bytes).  ascii must hold off[2*count] bytes; call with ascii == NULL to size it.
int vs_fastq_gather(const vs_fastq *fq, uint64_t first, uint64_t count, uint64_t *off, uint8_t *ascii) {
    if (!fq || !off || first + count > fq->n_pairs) return VS_E_ARG;
    off[0] = 0;
    for (uint64_t r = 0; r < count; r++) {
        off[2 * r + 1] = off[2 * r] + fq->f[0].seq_len[first + r];
        off[2 * r + 2] = off[2 * r + 1] + fq->f[1].seq_len[first + r];
    }
    if (!ascii) return VS_OK;
    const unsigned T = n_threads();
    parallel_for(T, [&](unsigned p) {
        uint64_t lo = count * p / T, hi = count * (p + 1) / T;
        for (uint64_t r = lo; r < hi; r++) {
            for (int w = 0; w < 2; w++) {
                const FqFile &f = fq->f[w];
                const uint32_t l = f.seq_len[first + r];
                if (l) memcpy(ascii + off[2 * r + w], f.text() + f.seq_start[first + r], l);
            }
        }
    });
    return VS_OK;
}

// Pairs [first, first+count) of the two files as a device read block (replaces
// PE_Inference.py:158-159 for that range).
int vs_fastq_block(vs_ctx *ctx, vs_fastq *fq, uint64_t first, uint64_t count, vs_reads **out) {
    if (!ctx || !fq || !out) return vs_fail(ctx, VS_E_ARG, "vs_fastq_block: bad argument");
    if (first + count > fq->n_pairs) return vs_fail(ctx, VS_E_RANGE, "vs_fastq_block: pairs %llu..%llu of %llu", (unsigned long long)first, (unsigned long long)(first + count), (unsigned long long)fq->n_pairs);
    VS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t need_off = sizeof(uint64_t) * (2 * count + 1);
    if (fq->stage_off_cap < need_off) {
        if (fq->stage_off) VS_HIP(ctx, hipHostFree(fq->stage_off));
        fq->stage_off = nullptr;
        fq->stage_off_cap = 0;
        VS_HIP(ctx, hipHostMalloc((void **)&fq->stage_off, need_off + need_off / 4, hipHostMallocDefault));
        fq->stage_off_cap = need_off + need_off / 4;
    }
    int rc = vs_fastq_gather(fq, first, count, fq->stage_off, nullptr);
    if (rc) return vs_fail(ctx, rc, "vs_fastq_block: gather failed");
    const size_t need = fq->stage_off[2 * count] ? fq->stage_off[2 * count] : 1;
    if (fq->stage_cap < need) {
        if (fq->stage) VS_HIP(ctx, hipHostFree(fq->stage));
        fq->stage = nullptr;
        fq->stage_cap = 0;
        VS_HIP(ctx, hipHostMalloc((void **)&fq->stage, need + need / 4, hipHostMallocDefault));
        fq->stage_cap = need + need / 4;
    }
    rc = vs_fastq_gather(fq, first, count, fq->stage_off, fq->stage);
    if (rc) return vs_fail(ctx, rc, "vs_fastq_block: gather failed");
    return vs_reads_pack(ctx, fq->stage, fq->stage_off, 2 * count, out);
}

}  // extern "C"

// ---- pe_info / st_info text -------------------------------------------------------------------
// utils/VStrains_PE_Inference.py:194-205 writes "{id_i}:{id_j}:{count}\n" for all i, j in
// row-major order, zeros included: N^2 lines (16.7 M at 4 k nodes).  Rows are sized and formatted
// on all host cores into one buffer, then written with a single write().
namespace {
inline uint32_t dec_digits(uint64_t v) {
    uint32_t d = 1;
    while (v >= 10) { v /= 10; d++; }
    return d;
}
inline char *put_dec(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
}  // namespace

extern "C" int vs_write_matrix_text(vs_ctx *ctx, const char *path, const uint8_t *ids, const uint64_t *id_off, uint32_t n,
                                    const int64_t *mat) {
    if (!path || !id_off || (n && (!ids || !mat))) return vs_fail(ctx, VS_E_ARG, "vs_write_matrix_text: bad argument");
    for (uint64_t i = 0; i < (uint64_t)n * n; i++)
        if (mat[i] < 0) return vs_fail(ctx, VS_E_ARG, "vs_write_matrix_text: negative count");
    const unsigned T = n_threads();
    std::vector<uint64_t> row_off((size_t)n + 1, 0);
    uint64_t ids_total = id_off[n] - id_off[0];
    parallel_for(T, [&](unsigned p) {
        uint32_t lo = (uint32_t)((uint64_t)n * p / T), hi = (uint32_t)((uint64_t)n * (p + 1) / T);
        for (uint32_t i = lo; i < hi; i++) {
            uint64_t li = id_off[i + 1] - id_off[i];
            uint64_t bytes = (uint64_t)n * (li + 3) + ids_total;  // id_i ':' id_j ':' ... '\n'
            const int64_t *row = mat + (uint64_t)i * n;
            for (uint32_t j = 0; j < n; j++) bytes += dec_digits((uint64_t)row[j]);
            row_off[i + 1] = bytes;
        }
    });
    for (uint32_t i = 0; i < n; i++) row_off[i + 1] += row_off[i];
    const uint64_t total = row_off[n];
    char *buf = (char *)malloc(total ? total : 1);
    if (!buf) return vs_fail(ctx, VS_E_OOM, "vs_write_matrix_text: %llu bytes", (unsigned long long)total);
    parallel_for(T, [&](unsigned p) {
        uint32_t lo = (uint32_t)((uint64_t)n * p / T), hi = (uint32_t)((uint64_t)n * (p + 1) / T);
        for (uint32_t i = lo; i < hi; i++) {
            char *q = buf + row_off[i];
            const uint8_t *idi = ids + id_off[i];
            const uint64_t li = id_off[i + 1] - id_off[i];
            const int64_t *row = mat + (uint64_t)i * n;
            for (uint32_t j = 0; j < n; j++) {
                memcpy(q, idi, li); q += li;
                *q++ = ':';
                const uint64_t lj = id_off[j + 1] - id_off[j];
                memcpy(q, ids + id_off[j], lj); q += lj;
                *q++ = ':';
                q = put_dec(q, (uint64_t)row[j]);
                *q++ = '\n';
            }
        }
    });
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        free(buf);
        return vs_fail(ctx, VS_E_ARG, "cannot open %s for writing: %s", path, strerror(errno));
    }
    uint64_t done = 0;
    while (done < total) {
        ssize_t w = write(fd, buf + done, (size_t)std::min<uint64_t>(total - done, 1u << 30));
        if (w < 0) {
            int e = errno;
            close(fd);
            free(buf);
            return vs_fail(ctx, VS_E_ARG, "write to %s failed: %s", path, strerror(e));
        }
        done += (uint64_t)w;
    }
    close(fd);
    free(buf);
    return VS_OK;
}
