// Native label-map writer: the per-image `Image.fromarray(label).save(png)` of the label loop (uest_seg_multi_os.py:929-931)
// as a pool of host threads fed with pinned staging buffers.  Host code only (no kernels): PNG = signature + IHDR + one IDAT
// (zlib deflate of the Up-filtered rows: class-id maps are piecewise constant, so "row minus the row above" is mostly zeros)
// + IEND.  The Python layer (mspl_amd/io.py) starts the device->host copy on a side stream, records an event and hands
// (buffer, event, paths) over; worker threads wait for the event, encode and write one image each, outside the interpreter
// -- the first version did this in Python threads and the label loop then spent 0.6 ms per batch fighting them for the GIL.
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {

struct Batch {
    const uint8_t* host;
    int n, H, W;
    hipEvent_t event;
    std::vector<std::string> paths;
    std::atomic<int> left;
    std::atomic<int> error;
    bool waited = false;            // event already synchronised (guarded by ev_mu)
    std::mutex ev_mu;
};

struct Task { Batch* b; int i; };

struct Writer {
    int level;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_task, cv_done;
    std::deque<Task> tasks;
    std::map<int64_t, Batch*> batches;      // ticket -> batch (erased by wait/done once finished)
    int64_t next_ticket = 0;
    bool stop = false;
};

void put_u32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void put_chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* body, size_t n) {
    put_u32(out, (uint32_t)n);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), body, body + n);
    put_u32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}

// 8-bit single-channel PNG of one (H, W) map
int encode_png(const uint8_t* img, int H, int W, int level, std::vector<uint8_t>& out) {
    std::vector<uint8_t> raw((size_t)H * (W + 1));
    for (int y = 0; y < H; ++y) {
        uint8_t* r = raw.data() + (size_t)y * (W + 1);
        const uint8_t* cur = img + (size_t)y * W;
        r[0] = 2;                                                        // filter type "Up"
        if (y == 0) {
            memcpy(r + 1, cur, (size_t)W);
        } else {
            const uint8_t* up = cur - W;
            for (int x = 0; x < W; ++x) r[1 + x] = (uint8_t)(cur[x] - up[x]);
        }
    }
    // Z_RLE: after the "Up" filter a label map is long runs of zeros with isolated boundary bytes; run-length matching finds all
    // of that structure at a fraction of the default strategy's hash-chain search time (zlib.h recommends it for PNG data).  The
    // stream is ordinary deflate: any PNG reader decodes it.
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15, 8, Z_RLE) != Z_OK) return -1;
    uLongf cap = deflateBound(&zs, (uLong)raw.size());
    std::vector<uint8_t> z(cap);
    zs.next_in = raw.data();  zs.avail_in = (uInt)raw.size();
    zs.next_out = z.data();  zs.avail_out = (uInt)cap;
    const int zrc = deflate(&zs, Z_FINISH);
    cap = zs.total_out;
    deflateEnd(&zs);
    if (zrc != Z_STREAM_END) return -1;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    out.clear();
    out.reserve(cap + 64);
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, (uint32_t)W); put_u32(ihdr, (uint32_t)H);
    const uint8_t tail[5] = {8, 0, 0, 0, 0};                             // bit depth 8, grey, deflate, adaptive, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(out, "IDAT", z.data(), cap);
    put_chunk(out, "IEND", nullptr, 0);
    return 0;
}

void worker_main(Writer* w) {
    std::vector<uint8_t> png;
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv_task.wait(lk, [&] { return w->stop || !w->tasks.empty(); });
            if (w->tasks.empty()) return;                                // stop requested and queue drained
            t = w->tasks.front();
            w->tasks.pop_front();
        }
        Batch* b = t.b;
        if (b->event) {                                                  // the D2H copy of this batch must have landed
            std::lock_guard<std::mutex> g(b->ev_mu);
            if (!b->waited) {
                if (hipEventSynchronize(b->event) != hipSuccess) b->error = 1;
                b->waited = true;
            }
        }
        if (!b->error) {
            const uint8_t* img = b->host + (size_t)t.i * b->H * b->W;
            if (encode_png(img, b->H, b->W, w->level, png) != 0) {
                b->error = 2;
            } else {
                FILE* f = fopen(b->paths[t.i].c_str(), "wb");
                if (!f || fwrite(png.data(), 1, png.size(), f) != png.size()) b->error = 3;
                if (f && fclose(f) != 0) b->error = 3;
            }
        }
        if (--b->left == 0) {
            std::lock_guard<std::mutex> lk(w->mu);
            w->cv_done.notify_all();
        }
    }
}

}  // namespace

extern "C" void* mspl_png_writer_create(int32_t workers, int32_t level) {
    if (workers < 1 || level < 0 || level > 9) return nullptr;
    Writer* w = new Writer();
    w->level = level;
    for (int i = 0; i < workers; ++i) w->threads.emplace_back(worker_main, w);
    return w;
}

extern "C" int64_t mspl_png_writer_submit(void* handle, const uint8_t* host, int32_t n, int32_t H, int32_t W,
                                          const char* const* paths, void* event) {
    MSPL_REQUIRE(handle && host && paths, MSPL_ERR_NULL_POINTER, "png_writer_submit: null pointer");
    MSPL_REQUIRE(n > 0 && H > 0 && W > 0, MSPL_ERR_BAD_SHAPE, "png_writer_submit: bad shape n=%d %dx%d", n, H, W);
    Writer* w = (Writer*)handle;
    Batch* b = new Batch();
    b->host = host; b->n = n; b->H = H; b->W = W; b->event = (hipEvent_t)event;
    b->left = n; b->error = 0;
    for (int i = 0; i < n; ++i) {
        MSPL_REQUIRE(paths[i], MSPL_ERR_NULL_POINTER, "png_writer_submit: path %d is NULL", i);
        b->paths.emplace_back(paths[i]);
    }
    std::lock_guard<std::mutex> lk(w->mu);
    const int64_t ticket = w->next_ticket++;
    w->batches[ticket] = b;
    for (int i = 0; i < n; ++i) w->tasks.push_back(Task{b, i});
    w->cv_task.notify_all();
    return ticket;
}

// 1: written (the staging buffer may be reused), 0: pending, negative: a file of the batch could not be written.
// block != 0 waits for completion.  A finished ticket is forgotten after it has been reported once.
extern "C" int mspl_png_writer_poll(void* handle, int64_t ticket, int32_t block) {
    MSPL_REQUIRE(handle, MSPL_ERR_NULL_POINTER, "png_writer_poll: null handle");
    Writer* w = (Writer*)handle;
    std::unique_lock<std::mutex> lk(w->mu);
    auto it = w->batches.find(ticket);
    MSPL_REQUIRE(it != w->batches.end(), MSPL_ERR_BAD_SHAPE, "png_writer_poll: unknown ticket %lld", (long long)ticket);
    Batch* b = it->second;
    if (block) w->cv_done.wait(lk, [&] { return b->left.load() == 0; });
    if (b->left.load() != 0) return 0;
    const int err = b->error.load();
    w->batches.erase(it);
    delete b;
    MSPL_REQUIRE(err == 0, MSPL_ERR_HIP, "png_writer: batch %lld failed (%s)", (long long)ticket,
                 err == 1 ? "event wait" : (err == 2 ? "deflate" : "file write"));
    return 1;
}

extern "C" int mspl_png_writer_destroy(void* handle) {
    if (!handle) return MSPL_OK;
    Writer* w = (Writer*)handle;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
        w->cv_task.notify_all();
    }
    for (auto& t : w->threads) t.join();                                 // workers drain the queue before leaving
    for (auto& kv : w->batches) delete kv.second;
    delete w;
    return MSPL_OK;
}
