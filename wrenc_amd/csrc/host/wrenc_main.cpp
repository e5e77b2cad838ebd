// wrenc_main.cpp -- the encoder as a native program with the reference's command line (main.rs:85-115).
//
//   wrenc -i in.yuv -o out.vvc --input-size 1920x1088 --output-size 1920x1088 --num-pictures 30
//         --qp 32 --max-split-depth 2 [--reconst rec.yuv]
//
// Flow of main.rs:117-402 over the two C ABIs of this repository (include/wrenc_gpu.h for the search and
// the final pass on the MI355X, include/wrenc_bitstream.h for everything written to the stream): VPS, SPS,
// PPS once; then pictures in batches (they are independent IDR pictures, main.rs:296): read Y, Cb, Cr at
// the output size, upload, search, read the record back, write picture header NAL + slice NAL, and the
// reconstruction when asked for.  Two sets of device slots and of page-locked host buffers alternate, so
// the GPU searches batch k+1 while batch k is read back and entropy coded on a pool of host threads.
// `-` is stdin / stdout.  Argument and I/O errors print `error: ...` on stderr and exit with status 0, as the
// reference does (main.rs:127-133); a failure inside the search or the stream writer (a HIP error, a level
// that overflows the rate tables, ...) is where the reference panics (block_splitter.rs:453): status 101,
// Rust's panic status, so that a truncated stream never comes with a success status.  Options the reference does not have: --batch, --threads, --device,
// --devices (several GPUs of the node, batches in turn), --verbose, --tokens auto|on|off (how a batch comes back: as the
// residual tokens the device makes of it, wrenc_gpu_download_tokens -- the host then runs the CU-level syntax and the
// arithmetic coder only, 1.8x less host time per picture, 20x the bytes over PCIe -- or as the compact level record with
// residual_coding on the host as until round 3; same bytes either way; auto = tokens while the host threads are what the
// run waits for).  Links only against the two C ABIs: no HIP, no Python.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../../include/wrenc_bitstream.h"
#include "../../../include/wrenc_gpu.h"

namespace {

[[noreturn]] void die(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fputs("error: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    exit(0); // main.rs:132: process::exit(0) on argument and I/O errors
}

// failures of the two libraries: the reference panics there (exit status 101)
[[noreturn]] void fatal(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fputs("error: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    fflush(nullptr);
    _exit(101);
}

bool parse_size(const char* text, int& w, int& h) {
    char tail = 0;
    return sscanf(text, "%dx%d%c", &w, &h, &tail) == 2 && w > 0 && h > 0;
}

bool read_exact(FILE* f, uint8_t* dst, size_t n) {
    size_t got = 0;
    while (got < n) {
        const size_t r = fread(dst + got, 1, n - got, f);
        if (r == 0) return false;
        got += r;
    }
    return true;
}

// A fixed set of worker threads running index ranges (the slices of one batch).
class Pool {
public:
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) threads_.emplace_back([this] { work(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : threads_) t.join();
    }
    // run fn(0..count-1) on the workers; returns at once
    void start(int count, std::function<void(int)> fn) {
        std::lock_guard<std::mutex> g(m_);
        fn_ = std::move(fn);
        count_ = count;
        next_ = 0;
        done_ = 0;
        cv_.notify_all();
    }
    // more indices for the function given to start(): count .. count + more - 1
    void extend(int more) {
        std::lock_guard<std::mutex> g(m_);
        count_ += more;
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(m_);
        idle_.wait(g, [this] { return done_ == count_; });
    }

private:
    void work() {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return stop_ || next_ < count_; });
            if (stop_) return;
            const int i = next_++;
            g.unlock();
            fn_(i);
            g.lock();
            if (++done_ == count_) idle_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, idle_;
    std::function<void(int)> fn_;
    int count_ = 0, next_ = 0, done_ = 0;
    bool stop_ = false;
};

struct HostSet { // one (device, slot set) unit: page-locked planes of one batch
    wrenc_gpu_ctx* ctx = nullptr;
    int base = 0;               // first slot of the set in its context
    uint8_t* in = nullptr;      // batch x (Y | Cb | Cr)
    uint8_t* rec = nullptr;     // batch x (Y | Cb | Cr), only with --reconst
    int16_t* lev = nullptr;     // batch x room for every 4x4 block of levels; the compact read-back fills the coded ones
    uint32_t* mask = nullptr;   // batch x mask of coded 4x4 blocks (wrenc_gpu_download_compact)
    std::vector<wrenc_gpu_compact> cps;
    uint32_t* tok_pool = nullptr;   // token read-back (wrenc_gpu_download_tokens): the pages of the batch
    size_t tok_cap = 0, tok_used = 0;
    uint32_t* tok_first = nullptr;  // batch x CTUs: first page of every CTU
    std::vector<wrenc_gpu_tokens> tks;
    bool bs_tokens = false; // the batch being written was read back as tokens
    std::atomic<long long> busy_ns{0}; // worker time spent on the batch being written
    uint8_t* maps = nullptr;    // batch x (cu_log2_size | luma_mode | chroma_mode)
    std::vector<std::vector<uint8_t>> nal; // per picture
    std::vector<int> status;
    std::vector<size_t> len;
    int count = 0, first_poc = 0;       // the batch being searched / read back in this set
    int bs_count = 0, bs_first_poc = 0; // the batch whose slices are being written from this set
};

} // namespace

int main(int argc, char** argv) {
    // 4 encode lanes + 1 copy stream per context: more hardware queues than the HIP runtime's default of 4 let the
    // read-back overlap the search (must be in the environment before the first HIP call; an explicit setting wins)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const char *input = nullptr, *output = nullptr, *reconst = nullptr, *in_size = nullptr, *out_size = nullptr,
               *extra = nullptr;
    long num_pictures = -1;
    int qp = 26; // ctu.rs:382 when --qp is absent
    int depth = 3, batch = 64, n_threads = 8, device = 0;
    bool verbose = false, use_tokens = true;
    int tokens_mode = 0; // --tokens auto (0) | on (1) | off (2)
    int ramp_mode = 0;   // --ramp-down auto (0) | always (1) | never (2)
    const char* device_list = nullptr;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const auto val = [&]() -> const char* {
            if (i + 1 >= argc) die("option %s needs a value", a.c_str());
            return argv[++i];
        };
        if (a == "-i" || a == "--input") input = val();
        else if (a == "-o" || a == "--output") output = val();
        else if (a == "-r" || a == "--reconst") reconst = val();
        else if (a == "--input-size") in_size = val();
        else if (a == "--output-size") out_size = val();
        else if (a == "--num-pictures") num_pictures = atol(val());
        else if (a == "--qp") qp = atoi(val());
        else if (a == "--max-split-depth") depth = atoi(val());
        else if (a == "--extra-params") extra = val();
        else if (a == "--batch") batch = atoi(val());
        else if (a == "--threads") n_threads = atoi(val());
        else if (a == "--device") device = atoi(val());
        else if (a == "--devices") device_list = val();
        else if (a == "--verbose") verbose = true;
        else if (a == "--no-tokens") tokens_mode = 2;
        else if (a == "--ramp-down") { // how a run ends: auto (smaller last batches when the host's tail is heavy), always, never
            const std::string v = val();
            ramp_mode = v == "always" ? 1 : (v == "never" ? 2 : (v == "auto" ? 0 : -1));
            if (ramp_mode < 0) die("Invalid ramp-down: %s (auto, always, never)", v.c_str());
        }
        else if (a == "--tokens") {
            const std::string v = val();
            tokens_mode = v == "on" ? 1 : (v == "off" ? 2 : (v == "auto" ? 0 : -1));
            if (tokens_mode < 0) die("Invalid tokens: %s (auto, on, off)", v.c_str());
        }
        else die("unknown option %s", a.c_str());
    }
    if (!input || !output || !in_size || !out_size || num_pictures < 0)
        die("the following options are required: --input --output --input-size --output-size --num-pictures");
    int w = 0, h = 0, iw = 0, ih = 0;
    if (!parse_size(in_size, iw, ih)) die("Invalid input-size: %s", in_size); // parsed, otherwise unused (main.rs:164-174)
    if (!parse_size(out_size, w, h)) die("Invalid output-size: %s", out_size);
    if (extra) {
        std::string e = extra;
        size_t pos = 0;
        while (pos <= e.size()) {
            const size_t end = e.find(',', pos) == std::string::npos ? e.size() : e.find(',', pos);
            const std::string item = e.substr(pos, end - pos);
            if (item.find('=') == std::string::npos || item.find('=') != item.rfind('='))
                die("Invalid extra-params: %s", extra);
            pos = end + 1;
        }
    }
    use_tokens = tokens_mode != 2;
    if (w % 32 || h % 32) die("output-size must be a multiple of the 32x32 CTU (picture.rs:178-181): %dx%d", w, h);
    if (qp < 0 || qp > 63 || depth < 0 || depth > 3) die("qp must be 0..63, max-split-depth 0..3");
    if (batch < 1) batch = 1;
    if (num_pictures > 0 && batch > num_pictures) batch = (int)num_pictures;
    if (n_threads < 1) n_threads = 1;
    std::vector<int> devices;
    if (device_list) { // "0,1,2,3": HIP device ordinals, one context each (an ordinal may repeat)
        for (const char* p = device_list; *p;) {
            char* end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p || d < 0 || (*end && *end != ',')) die("Invalid devices: %s", device_list);
            devices.push_back((int)d);
            p = *end ? end + 1 : end;
        }
    }
    if (devices.empty()) devices.push_back(device);

    FILE* fin = strcmp(input, "-") ? fopen(input, "rb") : stdin;
    if (!fin) die("failed to open input file: %s", strerror(errno));
    FILE* fout = strcmp(output, "-") ? fopen(output, "wb") : stdout;
    if (!fout) die("failed to open output file: %s", strerror(errno));
    FILE* frec = nullptr;
    if (reconst && !(frec = fopen(reconst, "wb"))) die("failed to open reconst file: %s", strerror(errno));

    // One context per entry of --devices (default: --device).  Every context gets two sets of slots; a (device,
    // set) pair is a "unit", and batches go to the units in turn: d0/s0, d1/s0, .., d0/s1, d1/s1, .. so that
    // consecutive batches run on different GPUs and every GPU always has a batch queued behind the one it is
    // searching.  Pictures are independent IDRs (main.rs:296): no data moves between GPUs.
    const int n_dev = (int)devices.size();
    const int per_dev = num_pictures > (long)batch * n_dev ? 2 : 1;
    wrenc_gpu_config cfg;
    if (wrenc_gpu_default_config(&cfg, w, h, qp, depth)) fatal("%s", wrenc_gpu_last_error(nullptr));
    if (extra && wrenc_gpu_config_extra_params(&cfg, extra)) fatal(  // a value that is not a number: parse().unwrap() panics in the reference
       "%s", wrenc_gpu_last_error(nullptr));
    cfg.n_slots = per_dev * batch;
    std::vector<wrenc_gpu_ctx*> ctxs;
    for (int d : devices) {
        cfg.device = d;
        wrenc_gpu_ctx* ctx = nullptr;
        if (wrenc_gpu_create(&cfg, &ctx)) fatal("%s", wrenc_gpu_last_error(nullptr)); // no CPU path: fails without an MI355X
        ctxs.push_back(ctx);
    }

    const size_t ysz = (size_t)w * h, csz = ysz / 4, pic = ysz + 2 * csz;
    const size_t n4 = ysz / 16, n8 = ysz / 64, maps = 2 * n4 + n8;
    const size_t mask_words = wrenc_gpu_compact_mask_words(w, h), level_blocks = pic / 16;
    const size_t n_ctus = (size_t)(w / 32) * (h / 32);
    std::vector<HostSet> units((size_t)(per_dev * n_dev));
    for (size_t u = 0; u < units.size(); ++u) {
        HostSet& s = units[u];
        s.ctx = ctxs[u % (size_t)n_dev];
        s.base = (int)(u / (size_t)n_dev) * batch;
        s.in = (uint8_t*)wrenc_gpu_alloc_host(s.ctx, pic * batch);
        s.maps = (uint8_t*)wrenc_gpu_alloc_host(s.ctx, maps * batch);
        if (use_tokens) {
            // room for the batch's tokens: what textured content takes at this QP (4-byte tokens per luma sample: ~1 at QP 32,
            // ~3 at QP 22) with a margin; a batch that needs more is read back as the compact level record instead
            const double per_sample = qp >= 30 ? 1.5 : (qp >= 25 ? 2.5 : 4.5);
            s.tok_cap = (size_t)((double)ysz * batch * per_sample) / WRENC_GPU_TOKEN_PAGE * WRENC_GPU_TOKEN_PAGE + WRENC_GPU_TOKEN_PAGE * 1024;
            s.tok_pool = (uint32_t*)wrenc_gpu_alloc_host(s.ctx, s.tok_cap * sizeof(uint32_t));
            s.tok_first = (uint32_t*)wrenc_gpu_alloc_host(s.ctx, n_ctus * sizeof(uint32_t) * batch);
            s.tks.resize((size_t)batch);
            if (!s.tok_pool || !s.tok_first) fatal("%s", wrenc_gpu_last_error(s.ctx));
        } else {
            s.lev = (int16_t*)wrenc_gpu_alloc_host(s.ctx, pic * batch * sizeof(int16_t));
            s.mask = (uint32_t*)wrenc_gpu_alloc_host(s.ctx, mask_words * sizeof(uint32_t) * batch);
            if (!s.lev || !s.mask) fatal("%s", wrenc_gpu_last_error(s.ctx));
        }
        s.cps.resize((size_t)batch);
        if (frec) s.rec = (uint8_t*)wrenc_gpu_alloc_host(s.ctx, pic * batch);
        if (!s.in || !s.maps || (frec && !s.rec)) fatal("%s", wrenc_gpu_last_error(s.ctx));
        s.nal.resize((size_t)batch);
        s.status.assign((size_t)batch, 0);
        s.len.assign((size_t)batch, 0);
    }
    const auto gpu_check = [](HostSet& s, int rc) {
        if (rc) fatal("%s", wrenc_gpu_last_error(s.ctx));
    };

    {
        uint8_t hdr[512];
        size_t n = 0;
        if (wrenc_bs_write_parameter_sets(w, h, qp, hdr, sizeof(hdr), &n)) fatal("parameter sets do not fit");
        fwrite(hdr, 1, n, fout);
    }

    const auto t_start = std::chrono::steady_clock::now();
    long poc = 0, pictures = 0;
    unsigned long long bytes = 0;
    // read + upload the next batch into the unit's slots and start its search (asynchronous)
    // A regular input file is read by several threads at once (pread at picture offsets): one thread copies a
    // picture of 3 MB out of the page cache in about a millisecond, and the first batch's read is the one part of the
    // run that nothing overlaps.  A pipe is read in order by this thread.
    const bool seekable = fin != stdin && lseek(fileno(fin), 0, SEEK_CUR) != (off_t)-1;
    const int n_readers = seekable ? (n_threads < 8 ? n_threads : 8) : 1;
    bool tail_heavy = false; // writing a batch's slices keeps the threads busy for more than 0.4 of the batch's turn
    const auto submit = [&](HostSet& s) {
        s.count = 0;
        s.first_poc = (int)poc;
        // The last batch's read-back and entropy coding overlap with nothing.  Where that is a good part of a batch's turn
        // (tail_heavy: textured content), the run ends on smaller batches -- a half, a quarter, a quarter of --batch, each a
        // little slower to search -- so that what is left at the end is a quarter's work.
        const long left = num_pictures - poc;
        int want = (int)(left < batch ? left : batch);
        if (per_dev == 2 && ramp_mode != 2 && (tail_heavy || ramp_mode == 1) && left <= batch && left > batch / 4) want = (int)(left / 2 > batch / 4 ? left / 2 : batch / 4);
        if (want < 1 && left > 0) want = 1;
        if (seekable && want > 0) { // (always pread then: the FILE's own position is never used)
            // readers fill the pictures (striped), this thread uploads each one as soon as it is there
            std::vector<std::atomic<int>> ready((size_t)want);
            for (auto& r : ready) r.store(0);
            std::vector<std::thread> readers;
            for (int t = 0; t < n_readers; ++t)
                readers.emplace_back([&, t] {
                    for (int k = t; k < want; k += n_readers) {
                        uint8_t* p = s.in + pic * k;
                        size_t got = 0;
                        const off_t at = (off_t)((size_t)(poc + k) * pic);
                        while (got < pic) {
                            const ssize_t r = pread(fileno(fin), p + got, pic - got, at + (off_t)got);
                            if (r <= 0) break;
                            got += (size_t)r;
                        }
                        ready[(size_t)k].store(got == pic ? 1 : -1, std::memory_order_release);
                    }
                });
            int short_at = -1;
            for (int k = 0; k < want; ++k) {
                int st;
                while ((st = ready[(size_t)k].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
                if (st < 0) {
                    short_at = k;
                    break;
                }
                uint8_t* p = s.in + pic * k;
                gpu_check(s, wrenc_gpu_upload(s.ctx, s.base + k, p, p + ysz, p + ysz + csz, (size_t)w, (size_t)w / 2));
                ++s.count;
            }
            for (std::thread& t : readers) t.join();
            if (short_at >= 0) die("input ended after %ld of %ld pictures", poc + short_at, num_pictures);
        } else {
            for (int k = 0; k < want; ++k) {
                uint8_t* p = s.in + pic * k;
                if (!read_exact(fin, p, pic)) die("input ended after %ld of %ld pictures", poc + k, num_pictures);
                gpu_check(s, wrenc_gpu_upload(s.ctx, s.base + k, p, p + ysz, p + ysz + csz, (size_t)w, (size_t)w / 2));
                ++s.count;
            }
        }
        if (s.count) gpu_check(s, wrenc_gpu_encode(s.ctx, s.base, s.count));
        poc += s.count;
    };
    Pool pool(n_threads);
    const size_t first_guess = ysz / 2 + 65536;
    // slices of the unit's batch on the pool; flush() collects them in picture order
    // (started empty: the pictures are handed to the pool as their read-back completes, pool.extend)
    const auto start_slices = [&](HostSet& s) {
        s.bs_count = s.count;
        s.bs_first_poc = s.first_poc;
        s.busy_ns.store(0);
        pool.start(0, [&s, w, h, qp, pic, ysz, csz, maps, n4, first_guess, mask_words, n_ctus](int k) {
            struct Busy { // (every exit of the function adds its time)
                HostSet& s;
                std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                ~Busy() { s.busy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
            } busy{s};
            const uint8_t* m = s.maps + maps * k;
            if (s.bs_tokens) {
                // the device made the residual tokens: CU-level syntax + arithmetic coder here
                const wrenc_bs_tokens tk = {m, m + n4, m + 2 * n4, s.tok_pool, s.tok_cap, s.tok_first + n_ctus * k};
                std::vector<uint8_t>& out = s.nal[(size_t)k];
                if (out.size() < first_guess) out.resize(first_guess);
                size_t n = 0;
                int rc = wrenc_bs_write_picture_tokens(w, h, qp, s.bs_first_poc + k, &tk, out.data(), out.size(), &n);
                if (rc == WRENC_BS_ENOSPC) {
                    out.resize(n);
                    rc = wrenc_bs_write_picture_tokens(w, h, qp, s.bs_first_poc + k, &tk, out.data(), out.size(), &n);
                }
                s.status[(size_t)k] = rc;
                s.len[(size_t)k] = rc ? 0 : n;
                return;
            }
            // the level planes the stream writer reads, rebuilt from the compact record in this thread's own buffer
            static thread_local std::vector<int16_t> dense;
            if (dense.size() < pic) dense.resize(pic);
            int16_t* l = dense.data();
            wrenc_gpu_expand_levels(w, h, s.mask + mask_words * k, s.lev + pic * k, l, l + ysz, l + ysz + csz);
            const wrenc_bs_record rec = {m, m + n4, m + 2 * n4, l, l + ysz, l + ysz + csz};
            std::vector<uint8_t>& out = s.nal[(size_t)k];
            // wrenc_bs_picture_bound is the proven worst case (12 bytes per luma sample); real pictures need a
            // small fraction, and the writer reports the size it needs when the buffer is too small
            if (out.size() < first_guess) out.resize(first_guess);
            size_t n = 0;
            int rc = wrenc_bs_write_picture(w, h, qp, s.bs_first_poc + k, &rec, out.data(), out.size(), &n);
            if (rc == WRENC_BS_ENOSPC) {
                out.resize(n);
                rc = wrenc_bs_write_picture(w, h, qp, s.bs_first_poc + k, &rec, out.data(), out.size(), &n);
            }
            s.status[(size_t)k] = rc;
            s.len[(size_t)k] = rc ? 0 : n;
        });
    };
    const auto flush = [&](HostSet& s) {
        pool.wait();
        for (int k = 0; k < s.bs_count; ++k) {
            if (s.status[(size_t)k]) fatal("wrenc_bs_write_picture failed with %d on picture %d", s.status[(size_t)k], s.bs_first_poc + k);
            fwrite(s.nal[(size_t)k].data(), 1, s.len[(size_t)k], fout);
            bytes += s.len[(size_t)k];
            if (frec) fwrite(s.rec + pic * k, 1, pic, frec); // main.rs:387-399
        }
        pictures += s.bs_count;
    };

    // Fill every unit, then go round: read the oldest batch back (waits for its search only), collect the
    // slices of the batch before it (written meanwhile), start this batch's slices, and give the unit the next
    // batch.  With --reconst the planes of a batch are written out before its unit is read back into again.
    for (HostSet& s : units)
        if (poc < num_pictures) submit(s);
    HostSet* pending = nullptr;
    // --verbose: where the main thread spends the run (waiting for slices, for the search + read-back, reading + uploading)
    double t_flush = 0, t_readback = 0, t_submit = 0;
    int n_token_batches = 0;
    auto t_turn = std::chrono::steady_clock::now();
    const auto now = [] { return std::chrono::steady_clock::now(); };
    const auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    if (verbose) fprintf(stderr, "first batches submitted after %.3f s\n", since(t_start));
    for (size_t head = 0; units[head].count > 0; head = (head + 1) % units.size()) {
        HostSet& s = units[head];
        auto tp = now();
        // This batch is read back in ONE call (waits for its search) while the pool still writes the previous batch's
        // slices (other buffers): the residual tokens the device made of it, or -- when they do not fit, or with
        // --no-tokens -- the compact level record (mask of coded 4x4 blocks + those blocks); either way with the maps.
        bool tokens_done = false;
        // --tokens auto = tokens: the pass costs the device about 2 % of the search's time and the host writes a picture 1.8 ..
        // 2.1x faster from them; the compact level record is what a batch falls back to when its tokens do not fit the pool
        // (and what --tokens off reads)
        const bool want_tokens = use_tokens;
        if (want_tokens) {
            for (int k = 0; k < s.count; ++k) {
                uint8_t* m = s.maps + maps * k;
                uint8_t* r = frec ? s.rec + pic * k : nullptr;
                s.tks[(size_t)k] = wrenc_gpu_tokens{s.tok_first + n_ctus * k, m, m + n4, m + 2 * n4, r, r ? r + ysz : nullptr, r ? r + ysz + csz : nullptr};
            }
            const int rc = wrenc_gpu_download_tokens(s.ctx, s.base, s.count, s.tks.data(), s.tok_pool, s.tok_cap, &s.tok_used);
            if (rc == WRENC_GPU_OK)
                tokens_done = true;
            else if (rc != WRENC_GPU_ENOMEM)
                gpu_check(s, rc);
            else if (verbose)
                fprintf(stderr, "batch at picture %d: more tokens than the pool holds, read back as the compact level record\n", s.first_poc);
        }
        if (!tokens_done) {
            if (!s.lev) { // (first fallback of this set)
                s.lev = (int16_t*)wrenc_gpu_alloc_host(s.ctx, pic * batch * sizeof(int16_t));
                s.mask = (uint32_t*)wrenc_gpu_alloc_host(s.ctx, mask_words * sizeof(uint32_t) * batch);
                if (!s.lev || !s.mask) fatal("%s", wrenc_gpu_last_error(s.ctx));
            }
            for (int k = 0; k < s.count; ++k) {
                uint8_t* m = s.maps + maps * k;
                uint8_t* r = frec ? s.rec + pic * k : nullptr;
                s.cps[(size_t)k] = wrenc_gpu_compact{s.mask + mask_words * k, s.lev + pic * k, level_blocks, 0, m, m + n4, m + 2 * n4,
                                                     r, r ? r + ysz : nullptr, r ? r + ysz + csz : nullptr};
            }
            gpu_check(s, wrenc_gpu_download_compact(s.ctx, s.base, s.count, s.cps.data()));
        }
        t_readback += since(tp);
        if (verbose) fprintf(stderr, "  %.3f s: batch at picture %d read back (%s)", since(t_start), s.first_poc, tokens_done ? "tokens" : "compact");
        tp = now();
        if (pending) flush(*pending); // the previous batch's slices, in picture order, to the output
        if (verbose) fprintf(stderr, ", %.3f s: previous batch's slices out", since(t_start));
        {
            const double waited = since(tp), turn = since(t_turn);
            t_flush += waited;
            if (pending && pending->bs_count > 0 && turn > 0) tail_heavy = (double)pending->busy_ns.load() * 1e-9 > 0.4 * n_threads * turn;
            n_token_batches += tokens_done ? 1 : 0;
            t_turn = now();
        }
        tp = now();
        s.bs_tokens = tokens_done;
        start_slices(s);
        pool.extend(s.count);
        pending = &s;
        s.count = 0;
        if (poc < num_pictures) {
            if (units.size() == 1) { // a single unit: its slices must be out before its buffers are refilled
                flush(s);
                pending = nullptr;
            }
            submit(s);
        }
        t_submit += since(tp);
        if (verbose) fprintf(stderr, ", %.3f s: next batch submitted\n", since(t_start));
    }
    {
        const auto tp = now();
        if (pending) flush(*pending);
        t_flush += since(tp);
    }
    if (verbose)
        fprintf(stderr, "main thread: %.3f s waiting for slices, %.3f s for search + read-back, %.3f s reading + uploading; %d batch(es) read back as tokens\n",
                t_flush, t_readback, t_submit, n_token_batches);
    fflush(fout);
    if (frec) fclose(frec);
    if (fout != stdout) fclose(fout);
    if (verbose) {
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        fprintf(stderr, "%ld pictures, %llu bytes, %.2f s, %.1f pictures/s (file to stream, %d GPU context(s), %d host threads)\n",
                pictures, bytes, dt, pictures / (dt > 0 ? dt : 1e-9), n_dev, n_threads);
    }
    for (HostSet& s : units) {
        wrenc_gpu_free_host(s.ctx, s.in);
        wrenc_gpu_free_host(s.ctx, s.lev);
        wrenc_gpu_free_host(s.ctx, s.maps);
        wrenc_gpu_free_host(s.ctx, s.mask);
        wrenc_gpu_free_host(s.ctx, s.tok_pool);
        wrenc_gpu_free_host(s.ctx, s.tok_first);
        wrenc_gpu_free_host(s.ctx, s.rec);
    }
    for (wrenc_gpu_ctx* ctx : ctxs) wrenc_gpu_destroy(ctx);
    return 0;
}
