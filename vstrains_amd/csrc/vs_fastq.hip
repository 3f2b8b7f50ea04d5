// FASTQ ingest (host side, multi-threaded): the reference reads both files with readlines() in
// text mode, takes record r as lines 4r..4r+3 and uses line 4r+1 minus its LAST CHARACTER as the
// sequence (utils/VStrains_PE_Inference.py:146-159).  Text mode means universal newlines: "\r\n"
// and a lone "\r" both end a line.  A final line without a newline still loses its last (real)
// character.  total = min(len_f // 4, len_r // 4).
//
// vs_fastq_open maps both files and indexes the sequence lines in ONE pass over the text (every
// host core takes a byte range and notes where its newlines are; line numbers follow from the
// per-range counts).  vs_fastq_block packs the sequences of a record range to 2 bits per base ON
// THE HOST CORES, straight into pinned staging (a quarter of the ASCII bytes cross PCIe, no device
// packing pass), and enqueues the upload into device buffers recycled from earlier blocks; two
// staging sets alternate, so the cores can pack block i+1 while block i is uploaded and counted.
// No Python objects, no per-record allocation.
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "vs_internal.h"
#include "vs_pack_host.h"

namespace {

// per-record arrays: allocated without a fill (every entry is written by the part of the index
// pass that owns its line, which is also the thread that first touches the page)
template <typename T>
struct RawArray {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void alloc(size_t k) { p.reset(k ? new T[k] : nullptr); n = k; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

struct FqFile {
    const uint8_t *data = nullptr;
    size_t size = 0;
    int fd = -1;
    const uint8_t *map_base = nullptr;  // what munmap gets back (data / size may describe a byte range of the mapping)
    size_t map_size = 0;
    std::vector<uint8_t> inflated;     // only for a gzip file: its text (data / size then describe this copy)
    std::vector<uint8_t> translated;   // only when the file holds '\r': universal-newline copy
    RawArray<uint64_t> seq_start;      // per record: first byte of line 4r+1
    RawArray<uint32_t> seq_len;        // ... its length after dropping the last character
    uint64_t n_lines = 0;
    const uint8_t *text() const { return translated.empty() ? data : translated.data(); }
    size_t text_size() const { return translated.empty() ? size : translated.size(); }
};

// Host threads of the ingest: what this process may really run on -- the affinity mask cut by the
// cgroup CPU quota (a box that shows 256 CPUs under a 16-core quota throttles 64 busy threads at the
// next period boundary) -- unless VS_HOST_THREADS says otherwise.
unsigned n_threads() {
    static const unsigned machine = [] {
        unsigned n = std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
        if (FILE *fh = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char a[32] = {0};
            double period = 0.0;
            if (fscanf(fh, "%31s %lf", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0.0) {
                const double cores = atof(a) / period;
                if (cores >= 1.0 && cores < (double)n) n = (unsigned)(cores + 0.5);
            }
            fclose(fh);
        }
        return n;
    }();
    unsigned n = machine;
    if (const char *ev = getenv("VS_HOST_THREADS")) n = (unsigned)atoi(ev);
    return std::max(1u, std::min(n, 64u));
}

// Worker threads that outlive a call: the ingest runs a dozen short parallel loops per block, and
// starting 64 threads for each costs more than some of the loops.  Parts are claimed off a counter
// by the workers and by the caller; one loop at a time (callers queue on `turn`).  The pool is a
// leaked heap object with detached threads, replaced in a forked child (which has no workers).
class Pool {
  public:
    void run(unsigned parts, const std::function<void(unsigned)> &fn) {
        if (parts <= 1) {
            if (parts) fn(0u);
            return;
        }
        std::lock_guard<std::mutex> one_at_a_time(turn_);
        {
            std::lock_guard<std::mutex> lk(m_);
            while (workers_ + 1u < parts) {
                std::thread([this] { work(); }).detach();
                workers_++;
            }
            fn_ = &fn;
            parts_ = parts;
            next_ = 0;
            done_ = 0;
            gen_++;
        }
        cv_job_.notify_all();
        claim_and_run();
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return done_ == parts_; });
        fn_ = nullptr;
    }

  private:
    void claim_and_run() {
        for (;;) {
            const std::function<void(unsigned)> *fn;
            unsigned i;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!fn_ || next_ >= parts_) return;
                i = next_++;
                fn = fn_;
            }
            (*fn)(i);
            std::lock_guard<std::mutex> lk(m_);
            if (++done_ == parts_) cv_done_.notify_all();
        }
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_job_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
            }
            claim_and_run();
        }
    }
    std::mutex turn_, m_;
    std::condition_variable cv_job_, cv_done_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned parts_ = 0, next_ = 0, done_ = 0, workers_ = 0;
    uint64_t gen_ = 0;
};
Pool *g_pool = nullptr;
std::once_flag g_pool_once;

Pool &pool() {
    std::call_once(g_pool_once, [] {
        g_pool = new Pool();
        pthread_atfork(nullptr, nullptr, [] { g_pool = new Pool(); });  // (the child starts its own workers)
    });
    return *g_pool;
}

template <typename F>
void parallel_for(unsigned parts, F fn) {
    const std::function<void(unsigned)> f = fn;
    pool().run(parts, f);
}

// (errors go to `err`: the two files of a pair are mapped by two threads)
int map_file(std::string &err, const char *path, FqFile &f) {
    auto fail = [&](int code, const char *fmt, auto... args) {
        char buf[512];
        snprintf(buf, sizeof buf, fmt, args...);
        err = buf;
        return code;
    };
    f.fd = open(path, O_RDONLY);
    if (f.fd < 0) return fail(VS_E_ARG, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(f.fd, &st) != 0) return fail(VS_E_ARG, "cannot stat %s: %s", path, strerror(errno));
    f.size = (size_t)st.st_size;
    if (f.size) {
        void *p = mmap(nullptr, f.size, PROT_READ, MAP_PRIVATE, f.fd, 0);
        if (p == MAP_FAILED) return fail(VS_E_OOM, "cannot map %s: %s", path, strerror(errno));
        f.data = (const uint8_t *)p;
        f.map_base = f.data;
        f.map_size = f.size;
        madvise(p, f.size, MADV_SEQUENTIAL);
    }
    // A gzip file (magic 1f 8b; several members in a row as bgzip writes them are fine) is inflated
    // into memory once and then treated like the text it holds.  Beyond the reference, which opens
    // plain text only (SURVEY 8f-2 "optional"); one stream inflates on one core, the two files of a
    // pair side by side (vs_fastq_open).
    if (f.size >= 2 && f.data[0] == 0x1f && f.data[1] == 0x8b) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 15 + 16) != Z_OK) return fail(VS_E_OOM, "%s: zlib cannot start", path);
        std::vector<uint8_t> &out = f.inflated;
        out.resize(std::max<size_t>(f.size * 4, 1u << 16));
        size_t have = 0, in_at = 0;
        int rc = Z_OK;
        for (;;) {
            const size_t in_now = std::min<size_t>(f.size - in_at, 1u << 30);
            zs.next_in = const_cast<Bytef *>(f.data + in_at);
            zs.avail_in = (uInt)in_now;
            if (out.size() - have < (1u << 16)) out.resize(out.size() + out.size() / 2);
            const size_t out_now = std::min<size_t>(out.size() - have, 1u << 30);
            zs.next_out = out.data() + have;
            zs.avail_out = (uInt)out_now;
            rc = inflate(&zs, Z_NO_FLUSH);
            in_at += in_now - zs.avail_in;
            have += out_now - zs.avail_out;
            if (rc == Z_STREAM_END) {
                if (in_at >= f.size) break;
                if (inflateReset(&zs) != Z_OK) { rc = Z_DATA_ERROR; break; }  // next member
                continue;
            }
            if (rc == Z_BUF_ERROR && zs.avail_out == 0) continue;  // (output full: grown at the top of the loop)
            if (rc != Z_OK) break;
            if (in_at >= f.size && zs.avail_out != 0) { rc = Z_DATA_ERROR; break; }  // truncated stream
        }
        inflateEnd(&zs);
        if (rc != Z_STREAM_END) return fail(VS_E_ARG, "%s: not a complete gzip stream (zlib code %d)", path, rc);
        out.resize(have);
        munmap((void *)f.data, f.size);
        f.map_base = nullptr;
        f.map_size = 0;
        close(f.fd);
        f.fd = -1;
        f.data = out.data();
        f.size = have;
    }
    return VS_OK;
}

// Python's text mode decodes the WHOLE file (PE_Inference.py:147-152: open(...).readlines()): a byte that is not valid
// UTF-8 anywhere -- header, sequence or quality line -- raises UnicodeDecodeError before anything is counted.  Here: the
// characters that START in [lo, hi) of txt[0, size) are checked (a caller cuts a file into ranges: a character that
// straddles a cut belongs to the range it starts in; leading continuation bytes belong to the range before).  Words
// without a high bit are skipped eight bytes at a time.
inline uint32_t utf8_len_strict(const uint8_t *q, size_t n) {  // bytes of the character at q[0], 0 if invalid (no overlong forms, no surrogates, nothing above U+10FFFF)
    const uint8_t c = q[0];
    if (c < 0x80u) return 1u;
    auto cont = [&](size_t i) { return i < n && (q[i] & 0xC0u) == 0x80u; };
    if (c >= 0xC2u && c <= 0xDFu) return cont(1) ? 2u : 0u;
    if (c >= 0xE0u && c <= 0xEFu) {
        if (!cont(1) || !cont(2)) return 0u;
        if (c == 0xE0u && q[1] < 0xA0u) return 0u;
        if (c == 0xEDu && q[1] >= 0xA0u) return 0u;
        return 3u;
    }
    if (c >= 0xF0u && c <= 0xF4u) {
        if (!cont(1) || !cont(2) || !cont(3)) return 0u;
        if (c == 0xF0u && q[1] < 0x90u) return 0u;
        if (c == 0xF4u && q[1] >= 0x90u) return 0u;
        return 4u;
    }
    return 0u;
}
bool utf8_range_ok(const uint8_t *txt, size_t size, size_t lo, size_t hi) {
    size_t i = lo;
    if (lo > 0 && lo < hi && (txt[lo] & 0xC0u) == 0x80u) {
        // a continuation byte at the cut: the tail of a character that starts in the range before (which checks it), or a
        // stray one.  Its lead byte lies at most three bytes back and says how far the character reaches.
        size_t j = lo;
        while (j > 0 && lo - j < 3u && (txt[j - 1u] & 0xC0u) == 0x80u) j--;
        if (j == 0) return false;
        const uint8_t lead = txt[j - 1u];
        const size_t len = lead >= 0xF0u ? 4u : lead >= 0xE0u ? 3u : lead >= 0xC0u ? 2u : 1u;
        if (j - 1u + len <= lo) return false;  // (no character reaches over the cut: the byte stands alone)
        i = j - 1u + len < hi ? j - 1u + len : hi;
    }
    while (i < hi) {
        if (i + 8u <= hi) {
            uint64_t w;
            memcpy(&w, txt + i, 8);
            if (!(w & 0x8080808080808080ull)) { i += 8u; continue; }
        }
        if (txt[i] < 0x80u) { i++; continue; }
        const uint32_t n = utf8_len_strict(txt + i, size - i);
        if (!n) return false;
        i += n;
    }
    return true;
}
const char *const BAD_UTF8_FILE_MSG = "%s holds bytes that are not valid UTF-8 (the reference's text-mode read raises UnicodeDecodeError)";

// Index the sequence lines of one file.
int index_file(vs_ctx *ctx, const char *path, FqFile &f) {
    const unsigned T = n_threads();
    std::vector<std::vector<uint64_t>> pos(T);
    std::vector<uint64_t> nl(T + 1, 0);
    // One pass: every part notes the positions of its newlines, a piece at a time (a piece is first
    // searched for '\r', then for '\n', while it is still in cache).
    // (r6) ... and, on the file's own bytes, checked for valid UTF-8 in the same visit: the check was a pass of its own over the
    // whole text -- a third of the open's memory traffic.  (Universal-newline translation changes ASCII bytes only: the
    // translated view is valid exactly when the file is.)
    std::vector<int> bad(T, 0);
    auto scan = [&](const uint8_t *txt, size_t n, bool first_visit) {
        std::vector<int> cr(T, 0);
        parallel_for(T, [&](unsigned p) {
            size_t lo = n * p / T, hi = n * (p + 1) / T;
            std::vector<uint64_t> &v = pos[p];
            v.clear();
            v.reserve((hi - lo) / 64 + 16);
#ifdef MADV_POPULATE_READ
            // (r6) a mapped file: this part's pages are brought into the page table in ONE call instead of a fault per 4 KB page
            // as the scan reaches it -- 600 k faults for 2.5 GB of text were most of the open's time.  A kernel that does not
            // know the advice says EINVAL and the scan faults the pages in as before.
            if (first_visit && f.map_base && txt == f.map_base && hi > lo) {
                const size_t pg = 4096u, a = lo & ~(pg - 1u);
                (void)madvise((void *)(txt + a), hi - a, MADV_POPULATE_READ);
            }
#endif
            for (size_t c0 = lo; c0 < hi; c0 += (256u << 10)) {
                const size_t c1 = c0 + (256u << 10) < hi ? c0 + (256u << 10) : hi;
                if (first_visit && !cr[p] && memchr(txt + c0, '\r', c1 - c0)) cr[p] = 1;
                if (first_visit && !bad[p] && !utf8_range_ok(txt, n, c0, c1)) bad[p] = 1;
                const uint8_t *q = txt + c0, *end = txt + c1;
                while (q < end) {
                    const uint8_t *r = (const uint8_t *)memchr(q, '\n', (size_t)(end - q));
                    if (!r) break;
                    v.push_back((uint64_t)(r - txt));
                    q = r + 1;
                }
            }
            nl[p + 1] = v.size();
        });
        bool any = false;
        for (unsigned p = 0; p < T; p++) any |= cr[p] != 0;
        return any;
    };
    // '\r' anywhere means universal-newline translation (rare): make that view once, then index it
    {
        const bool any_cr = scan(f.data, f.size, true);
        if (any_cr) {
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
            scan(f.translated.data(), f.translated.size(), false);
        }
    }
    const uint8_t *txt = f.text();
    const size_t n = f.text_size();
    for (unsigned p = 0; p < T; p++)
        if (bad[p]) return vs_fail(ctx, VS_E_UTF8, BAD_UTF8_FILE_MSG, path);
    for (unsigned p = 0; p < T; p++) nl[p + 1] += nl[p];
    const uint64_t n_newlines = nl[T];
    const bool open_tail = n > 0 && txt[n - 1] != '\n';
    f.n_lines = n_newlines + (open_tail ? 1 : 0);
    const uint64_t n_rec = f.n_lines / 4;
    f.seq_start.alloc(n_rec);
    f.seq_len.alloc(n_rec);
    std::vector<uint64_t> too_long(T, UINT64_MAX);  // per part: first record whose sequence line does not fit
    // the newline that ends line L (0-based) is newline number L; a sequence line has L % 4 == 1; it
    // starts right after newline L-1 (possibly the last one of an earlier part)
    parallel_for(T, [&](unsigned p) {
        const std::vector<uint64_t> &v = pos[p];
        if (v.empty()) return;
        uint64_t prev_end = 0;  // one past the newline before this part's first one
        bool have_prev = false;
        for (unsigned pp = p; pp-- > 0;)
            if (!pos[pp].empty()) { prev_end = pos[pp].back() + 1; have_prev = true; break; }
        (void)have_prev;
        uint64_t line = nl[p];
        uint64_t start = prev_end;
        for (size_t i = 0; i < v.size(); i++, line++) {
            if ((line & 3u) == 1u && (line >> 2) < n_rec) {
                f.seq_start[line >> 2] = start;
                f.seq_len[line >> 2] = (uint32_t)(v[i] - start);  // the dropped char is the newline
                if (v[i] - start > VS_LEN_MASK && too_long[p] == UINT64_MAX) too_long[p] = line >> 2;
            }
            start = v[i] + 1;
        }
    });
    if (open_tail) {  // last line without newline: it loses a real character
        const uint64_t line = n_newlines;
        if ((line & 3u) == 1u && (line >> 2) < n_rec) {
            uint64_t start = 0;
            for (unsigned pp = T; pp-- > 0;)
                if (!pos[pp].empty()) { start = pos[pp].back() + 1; break; }
            f.seq_start[line >> 2] = start;
            f.seq_len[line >> 2] = (uint32_t)(n - start - 1);
        }
    }
    for (unsigned p = 0; p < T; p++)
        if (too_long[p] != UINT64_MAX)
            return vs_fail(ctx, VS_E_RANGE, "%s: record %llu has a %u-byte sequence line", path, (unsigned long long)too_long[p], f.seq_len[too_long[p]]);
    return VS_OK;
}

void close_file(FqFile &f) {
    if (f.map_base) munmap((void *)f.map_base, f.map_size);
    f.map_base = nullptr;
    f.inflated = std::vector<uint8_t>();
    if (f.fd >= 0) close(f.fd);
    f.data = nullptr;
    f.fd = -1;
}

}  // namespace

struct vs_fastq {
    FqFile f[2];
    uint64_t n_pairs = 0;
    uint64_t bytes_indexed = 0;  // text bytes this handle went through (both files)
};

namespace {
// Python's text mode decodes the file (UTF-8), so the reference works on CHARACTERS: a valid multi-byte
// sequence inside a sequence line is ONE character of the read -- it counts once toward the read length and,
// being no key of the table, makes every window over it miss (PE_Inference.py:147-152, :25-26) -- and the
// "last character" a line loses (:158-159) is a whole character.  Sequences that hold a byte >= 0x80 (rare)
// are therefore turned into one byte per character before they are packed: ASCII bytes as they are, every
// multi-byte character as '?' (any byte outside ACGTN would do).  Bytes that are not valid UTF-8 make the
// reference's readlines() raise UnicodeDecodeError; here that is VS_E_UTF8.
const char *const BAD_UTF8_MSG = "a sequence line holds non-ASCII bytes that are not valid UTF-8 (the reference's text-mode read raises UnicodeDecodeError)";
inline bool has_high_bit(const uint8_t *q, uint32_t len) {
    uint8_t acc = 0;
    for (uint32_t i = 0; i < len; i++) acc |= q[i];
    return (acc & 0x80u) != 0;
}
// length in bytes of the UTF-8 character that starts at q[0] (n bytes available), 0 if invalid (Python's strict
// decoder: no overlong forms, no surrogates, nothing above U+10FFFF)
inline uint32_t utf8_char_len(const uint8_t *q, size_t n) {
    const uint8_t c = q[0];
    if (c < 0x80u) return 1u;
    auto cont = [&](size_t i) { return i < n && (q[i] & 0xC0u) == 0x80u; };
    if (c >= 0xC2u && c <= 0xDFu) return cont(1) ? 2u : 0u;
    if (c >= 0xE0u && c <= 0xEFu) {
        if (!cont(1) || !cont(2)) return 0u;
        if (c == 0xE0u && q[1] < 0xA0u) return 0u;   // overlong
        if (c == 0xEDu && q[1] >= 0xA0u) return 0u;  // surrogates
        return 3u;
    }
    if (c >= 0xF0u && c <= 0xF4u) {
        if (!cont(1) || !cont(2) || !cont(3)) return 0u;
        if (c == 0xF0u && q[1] < 0x90u) return 0u;   // overlong
        if (c == 0xF4u && q[1] >= 0x90u) return 0u;  // > U+10FFFF
        return 4u;
    }
    return 0u;
}
// The characters of record `rec`'s sequence (one byte each) into dst (may be NULL: count only).  Returns
// false on invalid UTF-8.  (The character a sequence line loses is always its newline: a record is four whole
// lines, so a sequence line never ends the file.)
inline bool seq_chars(const FqFile &f, uint64_t rec, uint8_t *dst, uint32_t *n_chars) {
    const uint8_t *q = f.text() + f.seq_start[rec];
    const uint32_t l = f.seq_len[rec];
    if (!has_high_bit(q, l)) {
        if (dst && l) memcpy(dst, q, l);
        *n_chars = l;
        return true;
    }
    uint32_t n = 0;
    for (size_t i = 0; i < l;) {
        const uint32_t cl = utf8_char_len(q + i, l - i);
        if (!cl) return false;
        if (dst) dst[n] = cl == 1u ? q[i] : (uint8_t)'?';
        n++;
        i += cl;
    }
    *n_chars = n;
    return true;
}

// pack one sequence; returns the flags (VS_FLAG_N / VS_FLAG_INVALID, 0x80 for non-ASCII bytes): vs_pack_host.cpp
static_assert(VS_PACK_FLAG_N == VS_FLAG_N && VS_PACK_FLAG_INVALID == VS_FLAG_INVALID, "flag values");
inline uint32_t pack_sequence(const uint8_t *q, uint32_t len, uint32_t *out) { return vs_pack_sequence_host(q, len, out); }
}  // namespace

extern "C" {

int vs_pack_sequence(const uint8_t *seq, uint32_t len, uint32_t *words, uint32_t *flags, int plain) {
    if ((!seq && len) || !words || !flags) return VS_E_ARG;
    *flags = plain ? vs_pack_sequence_host_plain(seq, len, words) : vs_pack_sequence_host(seq, len, words);
    return VS_OK;
}

int vs_fastq_open(vs_ctx *ctx, const char *fwd_path, const char *rve_path, vs_fastq **out) {
    if (!fwd_path || !rve_path || !out) return vs_fail(ctx, VS_E_ARG, "vs_fastq_open: bad argument");
    *out = nullptr;
    vs_fastq *fq = new vs_fastq();
    const char *paths[2] = {fwd_path, rve_path};
    // (mapping -- and, for gzip files, inflating -- the two files side by side; errors are reported in file order)
    int map_rc[2] = {VS_OK, VS_OK};
    std::string map_err[2];
    {
        std::thread second([&] { map_rc[1] = map_file(map_err[1], paths[1], fq->f[1]); });
        map_rc[0] = map_file(map_err[0], paths[0], fq->f[0]);
        second.join();
    }
    for (int i = 0; i < 2; i++) {
        int rc = map_rc[i];
        if (rc != VS_OK) vs_fail(ctx, rc, "%s", map_err[i].c_str());
        if (rc == VS_OK) rc = index_file(ctx, paths[i], fq->f[i]);
        if (rc != VS_OK) {
            close_file(fq->f[0]);
            close_file(fq->f[1]);
            delete fq;
            return rc;
        }
    }
    fq->n_pairs = std::min(fq->f[0].seq_start.size(), fq->f[1].seq_start.size());  // PE_Inference.py:154
    fq->bytes_indexed = fq->f[0].text_size() + fq->f[1].text_size();
    *out = fq;
    return VS_OK;
}

// ---- cooperative open (one process per GPU): nobody reads a whole file -------------------------------------------------
// Step 1, every rank for its byte range `part` of `n_parts`: newlines in the range.  out[0] = newlines, out[1] = file
// size, out[2] = flags: bit 0 the range holds a '\r' (universal-newline translation needed), bit 1 gzip file (cannot be
// cut), bit 2 the file does not end in a newline (set by the part that holds the end).  Host only.
int vs_fastq_count_part(const char *path, uint32_t part, uint32_t n_parts, uint64_t out[3]) {
    if (!path || !out || !n_parts || part >= n_parts) return VS_E_ARG;
    out[0] = out[1] = out[2] = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return vs_fail(nullptr, VS_E_ARG, "cannot open %s: %s", path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return vs_fail(nullptr, VS_E_ARG, "cannot stat %s: %s", path, strerror(errno)); }
    const size_t size = (size_t)st.st_size;
    out[1] = size;
    if (!size) { close(fd); return VS_OK; }
    void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) { close(fd); return vs_fail(nullptr, VS_E_OOM, "cannot map %s: %s", path, strerror(errno)); }
    const uint8_t *txt = (const uint8_t *)p;
    if (size >= 2 && txt[0] == 0x1f && txt[1] == 0x8b) {
        out[2] |= 2u;
    } else {
        const size_t lo = size / n_parts * part + std::min<size_t>(part, size % n_parts);
        const size_t hi = size / n_parts * (part + 1u) + std::min<size_t>(part + 1u, size % n_parts);
        const unsigned T = n_threads();
        std::vector<uint64_t> cnt(T, 0);
        std::vector<int> cr(T, 0), bad(T, 0);
        parallel_for(T, [&](unsigned t) {
            const size_t a = lo + (hi - lo) * t / T, b = lo + (hi - lo) * (t + 1) / T;
            uint64_t n = 0;
            const uint8_t *q = txt + a, *end = txt + b;
            if (a < b && memchr(q, '\r', b - a)) cr[t] = 1;
            if (a < b && !utf8_range_ok(txt, size, a, b)) bad[t] = 1;  // (every rank its own range: together the whole file)
            while (q < end) {
                const uint8_t *r = (const uint8_t *)memchr(q, '\n', (size_t)(end - q));
                if (!r) break;
                n++;
                q = r + 1;
            }
            cnt[t] = n;
        });
        for (unsigned t = 0; t < T; t++) { out[0] += cnt[t]; if (cr[t]) out[2] |= 1u; }
        if (hi == size && txt[size - 1] != '\n') out[2] |= 4u;
        for (unsigned t = 0; t < T; t++)
            if (bad[t]) {
                munmap(p, size);
                close(fd);
                return vs_fail(nullptr, VS_E_UTF8, BAD_UTF8_FILE_MSG, path);
            }
    }
    munmap(p, size);
    close(fd);
    return VS_OK;
}

namespace {
// byte offset at which line `line` (0-based) starts, from the newline counts of the n_parts byte ranges of the file
size_t line_start(const FqFile &f, uint64_t line, const uint64_t *counts, uint32_t n_parts) {
    if (line == 0) return 0;
    uint64_t before = 0;
    for (uint32_t part = 0; part < n_parts; part++) {
        if (before + counts[part] >= line) {  // the line-th newline lies in this range
            const size_t lo = f.size / n_parts * part + std::min<size_t>(part, f.size % n_parts);
            const size_t hi = f.size / n_parts * (part + 1u) + std::min<size_t>(part + 1u, f.size % n_parts);
            uint64_t need = line - before;
            const uint8_t *q = f.data + lo, *end = f.data + hi;
            while (q < end) {
                const uint8_t *r = (const uint8_t *)memchr(q, '\n', (size_t)(end - q));
                if (!r) break;
                if (--need == 0) return (size_t)(r - f.data) + 1u;
                q = r + 1;
            }
            return f.size;  // (counts that do not describe this file)
        }
        before += counts[part];
    }
    return f.size;  // fewer lines than asked for: the range is empty
}
}  // namespace

// Step 2, after the ranks have exchanged their counts (counts_f / counts_r: newlines of the n_parts byte ranges of the
// two files): records [first, last) of the pair -- only their bytes are indexed.  Plain text files without '\r' (the
// caller falls back to vs_fastq_open otherwise).  The handle numbers its records from 0; info[0] = last - first.
int vs_fastq_open_records(vs_ctx *ctx, const char *fwd_path, const char *rve_path, uint32_t n_parts, const uint64_t *counts_f,
                          const uint64_t *counts_r, uint64_t first, uint64_t last, vs_fastq **out) {
    if (!fwd_path || !rve_path || !out || !counts_f || !counts_r || !n_parts || last < first) return vs_fail(ctx, VS_E_ARG, "vs_fastq_open_records: bad argument");
    *out = nullptr;
    vs_fastq *fq = new vs_fastq();
    const char *paths[2] = {fwd_path, rve_path};
    const uint64_t *counts[2] = {counts_f, counts_r};
    for (int i = 0; i < 2; i++) {
        std::string err;
        int rc = map_file(err, paths[i], fq->f[i]);
        if (rc != VS_OK) vs_fail(ctx, rc, "%s", err.c_str());
        if (rc == VS_OK && !fq->f[i].inflated.empty()) rc = vs_fail(ctx, VS_E_ARG, "%s: a gzip file cannot be opened by record range", paths[i]);
        if (rc == VS_OK) {
            FqFile &f = fq->f[i];
            const size_t b0 = line_start(f, 4u * first, counts[i], n_parts), b1 = line_start(f, 4u * last, counts[i], n_parts);
            f.data += b0;
            f.size = b1 > b0 ? b1 - b0 : 0;
            fq->bytes_indexed += f.size;
            rc = index_file(ctx, paths[i], f);
            if (rc == VS_OK && !f.translated.empty()) rc = vs_fail(ctx, VS_E_ARG, "%s: holds '\\r'; open the whole file instead", paths[i]);
        }
        if (rc != VS_OK) {
            close_file(fq->f[0]);
            close_file(fq->f[1]);
            delete fq;
            return rc;
        }
    }
    fq->n_pairs = std::min(fq->f[0].seq_start.size(), fq->f[1].seq_start.size());
    if (fq->n_pairs != last - first) {
        const uint64_t got = fq->n_pairs;
        close_file(fq->f[0]);
        close_file(fq->f[1]);
        delete fq;
        return vs_fail(ctx, VS_E_STATE, "vs_fastq_open_records: %llu records in the byte ranges, %llu expected (stale counts?)",
                       (unsigned long long)got, (unsigned long long)(last - first));
    }
    *out = fq;
    return VS_OK;
}

void vs_fastq_close(vs_fastq *fq) {
    if (!fq) return;
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

uint64_t vs_fastq_bytes_indexed(const vs_fastq *fq) { return fq ? fq->bytes_indexed : 0;
}

// Sequence of record `record` of file `which` (0 forward, 1 reverse) into buf (cap bytes); *len
// receives its length.  Host only (tests, small tools).
int vs_fastq_sequence(const vs_fastq *fq, int which, uint64_t record, uint8_t *buf, uint32_t cap, uint32_t *len) {
    if (!fq || which < 0 || which > 1 || !len) return VS_E_ARG;
    const FqFile &f = fq->f[which];
    if (record >= f.seq_start.size()) return VS_E_RANGE;
    if (!seq_chars(f, record, nullptr, len)) return vs_fail(nullptr, VS_E_UTF8, "%s", BAD_UTF8_MSG);
    if (buf && cap >= *len && *len) {
        uint32_t n = 0;
        seq_chars(f, record, buf, &n);
    }
    return VS_OK;
}

// Host gather only: interleaved ASCII + offsets of pairs [first, first+count) (tests / callers
// that want the bytes).  ascii must hold off[2*count] bytes; call with ascii == NULL to size it.
int vs_fastq_gather(const vs_fastq *fq, uint64_t first, uint64_t count, uint64_t *off, uint8_t *ascii) {
    if (!fq || !off || first + count > fq->n_pairs) return VS_E_ARG;
    // lengths in CHARACTERS (= bytes unless a sequence holds multi-byte UTF-8, see seq_chars)
    const unsigned T = n_threads();
    std::vector<int> bad(T, 0);
    parallel_for(T, [&](unsigned p) {
        uint64_t lo = count * p / T, hi = count * (p + 1) / T;
        for (uint64_t r = lo; r < hi; r++)
            for (int w = 0; w < 2; w++) {
                uint32_t n = 0;
                if (!seq_chars(fq->f[w], first + r, nullptr, &n)) bad[p] = 1;
                off[2 * r + w + 1] = n;
            }
    });
    for (unsigned p = 0; p < T; p++)
        if (bad[p]) return vs_fail(nullptr, VS_E_UTF8, "%s", BAD_UTF8_MSG);
    off[0] = 0;
    for (uint64_t e = 0; e < 2 * count; e++) off[e + 1] += off[e];
    if (!ascii) return VS_OK;
    parallel_for(T, [&](unsigned p) {
        uint64_t lo = count * p / T, hi = count * (p + 1) / T;
        for (uint64_t r = lo; r < hi; r++)
            for (int w = 0; w < 2; w++) {
                uint32_t n = 0;
                seq_chars(fq->f[w], first + r, ascii + off[2 * r + w], &n);
            }
    });
    return VS_OK;
}

// Pairs [first, first+count) of the two files as a device read block (replaces
// PE_Inference.py:158-159 for that range).  The uploads are enqueued on the ctx stream and not
// waited for: the block is ready for anything enqueued behind them (vs_pe_count does).
int vs_fastq_block(vs_ctx *ctx, vs_fastq *fq, uint64_t first, uint64_t count, vs_reads **out) {
    if (!ctx || !fq || !out) return vs_fail(ctx, VS_E_ARG, "vs_fastq_block: bad argument");
    if (first + count > fq->n_pairs) return vs_fail(ctx, VS_E_RANGE, "vs_fastq_block: pairs %llu..%llu of %llu", (unsigned long long)first, (unsigned long long)(first + count), (unsigned long long)fq->n_pairs);
    *out = nullptr;
    const uint64_t n_ends = 2 * count;
    if (n_ends > 0xFFFFFFF0ull) return vs_fail(ctx, VS_E_RANGE, "vs_fastq_block: split the input into blocks of < 2^31 pairs");
    VS_HIP(ctx, hipSetDevice(ctx->device));
    // (the two pinned staging sets belong to the context: pinning 100 MB costs as much as packing it,
    // and a process opens one FASTQ pair after another -- shards, retries -- on the same context)
    FqStage &st = ctx->fq_stage[ctx->fq_next];
    ctx->fq_next ^= 1u;
    if (!st.done) VS_HIP(ctx, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    if (st.in_flight) {  // the block that used this set two calls ago
        VS_HIP(ctx, hipEventSynchronize(st.done));
        st.in_flight = false;
    }
    const unsigned T = n_threads();
    // words per thread range -> word offset of every end
    if (st.ends_cap < n_ends + 1) {
        if (st.woff) VS_HIP(ctx, hipHostFree(st.woff));
        if (st.meta) VS_HIP(ctx, hipHostFree(st.meta));
        st.woff = st.meta = nullptr;
        st.ends_cap = 0;
        const size_t cap = n_ends + n_ends / 8 + 16;
        VS_HIP(ctx, hipHostMalloc((void **)&st.woff, sizeof(uint32_t) * cap, hipHostMallocDefault));
        VS_HIP(ctx, hipHostMalloc((void **)&st.meta, sizeof(uint32_t) * cap, hipHostMallocDefault));
        st.ends_cap = cap;
    }
    std::vector<uint64_t> part_words(T + 1, 0);
    std::vector<uint32_t> part_max(T, 0);
    parallel_for(T, [&](unsigned p) {
        uint64_t lo = count * p / T, hi = count * (p + 1) / T, wsum = 0;
        uint32_t mx = 0;
        for (uint64_t r = lo; r < hi; r++)
            for (int w2 = 0; w2 < 2; w2++) {
                const uint32_t l = fq->f[w2].seq_len[first + r];
                wsum += (l + 15u) >> 4;
                mx = l > mx ? l : mx;
            }
        part_words[p + 1] = wsum;
        part_max[p] = mx;
    });
    for (unsigned p = 0; p < T; p++) part_words[p + 1] += part_words[p];
    const uint64_t words = part_words[T];
    if (words > 0xFFFFFFF0ull) return vs_fail(ctx, VS_E_RANGE, "vs_fastq_block: block exceeds 2^32 packed words");
    if (st.words_cap < words + VS_PAD_WORDS) {
        if (st.words) VS_HIP(ctx, hipHostFree(st.words));
        st.words = nullptr;
        st.words_cap = 0;
        const size_t cap = words + words / 8 + VS_PAD_WORDS;
        VS_HIP(ctx, hipHostMalloc((void **)&st.words, sizeof(uint32_t) * cap, hipHostMallocDefault));
        st.words_cap = cap;
    }
    std::vector<uint32_t> part_flags(T, 0);
    parallel_for(T, [&](unsigned p) {
        uint64_t lo = count * p / T, hi = count * (p + 1) / T;
        uint64_t wat = part_words[p];
        uint32_t any = 0;
        for (uint64_t r = lo; r < hi; r++)
            for (int w2 = 0; w2 < 2; w2++) {
                const FqFile &f = fq->f[w2];
                const uint32_t l = f.seq_len[first + r];
                const uint32_t fl = pack_sequence(f.text() + f.seq_start[first + r], l, st.words + wat);
                st.woff[2 * r + w2] = (uint32_t)wat;
                st.meta[2 * r + w2] = l | (fl << 24);
                any |= fl;
                wat += (l + 15u) >> 4;
            }
        part_flags[p] = any;
    });
    st.woff[n_ends] = (uint32_t)words;
    for (uint32_t i = 0; i < VS_PAD_WORDS; i++) st.words[words + i] = 0u;
    uint32_t any = 0, maxlen = 0;
    for (unsigned p = 0; p < T; p++) { any |= part_flags[p]; maxlen = part_max[p] > maxlen ? part_max[p] : maxlen; }
    if (any & (VS_FLAG_INVALID | 0x80u)) {
        // some read holds a byte outside ACGTN (rare): the device packer also builds the validity
        // mask and the position lists -- take that path with the plain bytes of this block (one byte
        // per CHARACTER: a multi-byte UTF-8 character arrives as one '?', see seq_chars)
        std::vector<uint64_t> off(n_ends + 1);
        int rc = vs_fastq_gather(fq, first, count, off.data(), nullptr);
        if (rc) return vs_fail(ctx, rc, "%s", rc == VS_E_UTF8 ? BAD_UTF8_MSG : "vs_fastq_block: gather failed");
        std::vector<uint8_t> bytes(off[n_ends] ? off[n_ends] : 1);
        rc = vs_fastq_gather(fq, first, count, off.data(), bytes.data());
        if (rc) return vs_fail(ctx, rc, "vs_fastq_block: gather failed");
        return vs_reads_pack(ctx, bytes.data(), off.data(), n_ends, out);
    }
    vs_reads *r = new vs_reads();
    r->n_ends = n_ends;
    r->n_words = words;
    r->max_len = maxlen;
    r->cached = true;
    const size_t b_woff = sizeof(uint32_t) * (n_ends + 1), b_meta = sizeof(uint32_t) * (n_ends ? n_ends : 1);
    const size_t b_words = sizeof(uint32_t) * (words + VS_PAD_WORDS);
    r->d_woff = vs_cache_alloc(ctx, b_woff);
    r->d_meta = vs_cache_alloc(ctx, b_meta);
    r->d_words = vs_cache_alloc(ctx, b_words);
    r->bytes = b_woff + b_meta + b_words;
    if (!r->d_woff || !r->d_meta || !r->d_words) {
        vs_reads_free(ctx, r);
        return vs_fail(ctx, VS_E_OOM, "vs_fastq_block: device buffers for %llu ends", (unsigned long long)n_ends);
    }
    hipError_t e1 = hipMemcpyAsync(r->d_woff, st.woff, b_woff, hipMemcpyHostToDevice, ctx->stream);
    if (e1 == hipSuccess && n_ends) e1 = hipMemcpyAsync(r->d_meta, st.meta, sizeof(uint32_t) * n_ends, hipMemcpyHostToDevice, ctx->stream);
    if (e1 == hipSuccess) e1 = hipMemcpyAsync(r->d_words, st.words, b_words, hipMemcpyHostToDevice, ctx->stream);
    if (e1 == hipSuccess) e1 = hipEventRecord(st.done, ctx->stream);
    if (e1 != hipSuccess) {
        vs_reads_free(ctx, r);
        return vs_fail(ctx, VS_E_HIP, "vs_fastq_block: %s", hipGetErrorString(e1));
    }
    st.in_flight = true;
    *out = r;
    return VS_OK;
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
    // The text of a 50 k-node graph is 36 GB per file: rows are formatted and written in blocks of at most
    // ~256 MB (whole rows; one row alone may be larger), the buffer reused.
    uint64_t BLOCK = 256ull << 20;
    if (const char *ev = getenv("VS_TEXT_BLOCK")) BLOCK = std::max<uint64_t>(1u, (uint64_t)atoll(ev));  // (tests: many small blocks)
    uint64_t cap = 0;
    for (uint32_t i0 = 0; i0 < n;) {
        uint32_t i1 = i0 + 1u;
        while (i1 < n && row_off[i1 + 1] - row_off[i0] <= BLOCK) i1++;
        cap = std::max(cap, row_off[i1] - row_off[i0]);
        i0 = i1;
    }
    char *buf = (char *)malloc(cap ? cap : 1);
    if (!buf) return vs_fail(ctx, VS_E_OOM, "vs_write_matrix_text: %llu bytes", (unsigned long long)cap);
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        free(buf);
        return vs_fail(ctx, VS_E_ARG, "cannot open %s for writing: %s", path, strerror(errno));
    }
    for (uint32_t i0 = 0; i0 < n;) {
        uint32_t i1 = i0 + 1u;
        while (i1 < n && row_off[i1 + 1] - row_off[i0] <= BLOCK) i1++;
        const uint64_t base = row_off[i0], bytes = row_off[i1] - base;
        const uint32_t rows = i1 - i0;
        parallel_for(std::min<unsigned>(T, rows), [&](unsigned p) {
            const unsigned parts = std::min<unsigned>(T, rows);
            uint32_t lo = i0 + (uint32_t)((uint64_t)rows * p / parts), hi = i0 + (uint32_t)((uint64_t)rows * (p + 1) / parts);
            for (uint32_t i = lo; i < hi; i++) {
                char *q = buf + (row_off[i] - base);
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
        // the block goes out as parallel pwrite()s of slices (the copy into the page cache is the slow part)
        const unsigned wparts = (unsigned)std::min<uint64_t>(T, std::max<uint64_t>(1u, bytes >> 22));
        std::vector<int> werr(wparts, 0);
        parallel_for(wparts, [&](unsigned p) {
            uint64_t lo = bytes * p / wparts, hi = bytes * (p + 1) / wparts;
            while (lo < hi) {
                ssize_t w = pwrite(fd, buf + lo, (size_t)std::min<uint64_t>(hi - lo, 1u << 30), (off_t)(base + lo));
                if (w < 0 && errno == EINTR) continue;
                if (w < 0) { werr[p] = errno; return; }
                lo += (uint64_t)w;
            }
        });
        for (unsigned p = 0; p < wparts; p++)
            if (werr[p]) {
                close(fd);
                free(buf);
                return vs_fail(ctx, VS_E_ARG, "write to %s failed: %s", path, strerror(werr[p]));
            }
        i0 = i1;
    }
    close(fd);
    free(buf);
    return VS_OK;
}
