// sidecar.cpp -- flat mirror of the stored tables the GPU shards are rebuilt from.
//
// The reference's source of truth is one redb file with the tables ucfp/fingerprints/v1, ucfp/vectors/v1 and
// ucfp/catalog/v2, all keyed (tenant_id, record_id) (src/index/embedded/mod.rs:37-43) and written in one transaction
// per upsert (:157-227).  redb's page format belongs to a crate that is not in the tree, so the drop-in mirrors those
// three tables into an APPEND-ONLY LOG the Rust host writes right after its redb transaction commits -- the same rows,
// the same catalog JSON (serde_json of CatalogEntry, :93-116) -- and the GPU side reads at start-up
// (EmbeddedBackend::open, :104-125) to rebuild its shards, keyed by the catalog's `algorithm` tag.
//
//   file   := "UCFPSC1\0" entry*
//   entry  := u32 payload_len | u32 crc32(payload) | payload | pad to 8
//   payload:= u8 op (1 upsert, 2 delete) | 3 x 0 | u32 tenant | u64 record_id | u32 fp_len | u32 dim | u32 json_len |
//             u32 0 | fingerprint | dim x f32 | catalog JSON
//
// Replay applies the log in order, the last entry of a key wins (redb insert / remove semantics).  A torn tail (crash
// in the middle of an append) fails its CRC and is cut off on the next open.  Pure host code: no HIP here.
//
// Concurrency (open-file-description locks on two bytes of the header, so they work between threads and processes):
//   byte 0  "a writer is alive": write-locked, non-blocking, for the life of a ucfp_sidecar -- a second ucfp_sidecar_open
//           of the same log fails instead of cutting the file under the first (redb allows one writer per file too);
//   byte 1  "the tail may move": write-locked by a writer only while it validates and cuts the torn tail at open,
//           read-locked by ucfp_sidecar_snapshot_open while it maps and walks the file, so a snapshot never touches
//           pages a concurrent open is truncating away (afterwards it only reads entries in front of the cut).
// The writer's descriptor is O_APPEND: every entry lands at the end whatever else happened to the offset.

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ucfp_hip.h"

namespace ucfp {
int capi_fail(int code, const char* fmt, ...);
}
using ucfp::capi_fail;

namespace {

constexpr char kMagic[8] = {'U', 'C', 'F', 'P', 'S', 'C', '1', '\0'};
constexpr uint32_t kHead = 32;      // fixed part of a payload
constexpr uint8_t kUpsert = 1, kDelete = 2;

uint32_t crc32_of(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
            table[i] = c;
        }
    });
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 255u] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

struct Entry {          // a parsed log entry; pointers into the mapping
    uint8_t op;
    uint32_t tenant;
    uint64_t id;
    const uint8_t* fp;
    uint32_t fp_len;
    const float* emb;
    uint32_t dim;
    const char* json;
    uint32_t json_len;
};

// Walks the entries of a mapped log.  Returns the offset of the first byte that is not part of a whole, valid entry.
template <class F>
size_t walk(const uint8_t* base, size_t size, F&& each) {
    size_t pos = sizeof kMagic;
    while (pos + 8 <= size) {
        uint32_t len, crc;
        memcpy(&len, base + pos, 4);
        memcpy(&crc, base + pos + 4, 4);
        if (len < kHead || pos + 8 + (size_t)len > size) break;
        const uint8_t* p = base + pos + 8;
        if (crc32_of(p, len) != crc) break;
        Entry e;
        e.op = p[0];
        memcpy(&e.tenant, p + 4, 4);
        memcpy(&e.id, p + 8, 8);
        memcpy(&e.fp_len, p + 16, 4);
        memcpy(&e.dim, p + 20, 4);
        memcpy(&e.json_len, p + 24, 4);
        if ((e.op != kUpsert && e.op != kDelete) || (uint64_t)kHead + e.fp_len + (uint64_t)e.dim * 4 + e.json_len != len) break;
        e.fp = p + kHead;
        e.emb = reinterpret_cast<const float*>(p + kHead + e.fp_len);      // read with memcpy: not necessarily aligned
        e.json = reinterpret_cast<const char*>(p + kHead + e.fp_len + (size_t)e.dim * 4);
        each(e);
        pos += 8 + (((size_t)len + 7) & ~(size_t)7);
    }
    return pos > size ? size : pos;
}

// serde_json writes `"algorithm":"<tag>"`; tags are plain ASCII identifiers (src/modality/*.rs), so no escapes to undo.
bool json_algorithm(const char* json, uint32_t n, const char** tag, uint32_t* tag_len) {
    static const char key[] = "\"algorithm\":\"";
    const size_t kl = sizeof key - 1;
    for (uint32_t i = 0; i + kl <= n; i++) {
        if (memcmp(json + i, key, kl) != 0) continue;
        uint32_t j = i + (uint32_t)kl;
        const uint32_t s = j;
        while (j < n && json[j] != '"') j++;
        if (j >= n) return false;
        *tag = json + s;
        *tag_len = j - s;
        return true;
    }
    return false;
}

// OFD lock on one header byte; type F_WRLCK / F_RDLCK / F_UNLCK.  Returns 0 or -1 (errno).
int lock_byte(int fd, off_t byte, short type, bool wait) {
    struct flock fl;
    memset(&fl, 0, sizeof fl);
    fl.l_type = type;
    fl.l_whence = SEEK_SET;
    fl.l_start = byte;
    fl.l_len = 1;
    int r;
    do r = fcntl(fd, wait ? F_OFD_SETLKW : F_OFD_SETLK, &fl);
    while (r != 0 && errno == EINTR);
    return r;
}

struct KeyHash {
    size_t operator()(const std::pair<uint32_t, uint64_t>& k) const {
        uint64_t z = k.second * 0x9e3779b97f4a7c15ull + k.first;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        return (size_t)(z ^ (z >> 31));
    }
};

}  // namespace

struct ucfp_sidecar {
    int fd = -1;
    std::mutex mu;
    std::vector<uint8_t> buf;
    uint64_t entries = 0;
};

struct ucfp_sidecar_snapshot {
    int fd = -1;
    const uint8_t* base = nullptr;
    size_t size = 0;
    std::vector<Entry> rows;       // live rows, ascending (tenant, record_id): the order of a redb range scan
};

extern "C" {

int ucfp_sidecar_open(const char* path, ucfp_sidecar** out) {
    if (!path || !out) return capi_fail(UCFP_E_INVALID, "path/out is NULL");
    *out = nullptr;
    const int fd = open(path, O_RDWR | O_CREAT | O_APPEND | O_CLOEXEC, 0644);
    if (fd < 0) return capi_fail(UCFP_E_INDEX, "sidecar %s: %s", path, strerror(errno));
    if (lock_byte(fd, 0, F_WRLCK, false) != 0) {
        const int e = errno;
        close(fd);
        if (e == EAGAIN || e == EACCES)
            return capi_fail(UCFP_E_INDEX, "sidecar %s is open for writing elsewhere (one writer per log)", path);
        return capi_fail(UCFP_E_INDEX, "sidecar %s: lock: %s", path, strerror(e));
    }
    if (lock_byte(fd, 1, F_WRLCK, true) != 0) {       // snapshots being opened finish their walk first
        const int e = errno;
        close(fd);
        return capi_fail(UCFP_E_INDEX, "sidecar %s: lock: %s", path, strerror(e));
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return capi_fail(UCFP_E_INDEX, "sidecar %s: %s", path, strerror(errno));
    }
    uint64_t entries = 0;
    if (st.st_size == 0) {
        if (write(fd, kMagic, sizeof kMagic) != (ssize_t)sizeof kMagic) {
            close(fd);
            return capi_fail(UCFP_E_INDEX, "sidecar %s: cannot write the header", path);
        }
    } else {
        // validate what is there; cut a torn tail
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            close(fd);
            return capi_fail(UCFP_E_INDEX, "sidecar %s: mmap: %s", path, strerror(errno));
        }
        const uint8_t* base = static_cast<const uint8_t*>(m);
        if ((size_t)st.st_size < sizeof kMagic || memcmp(base, kMagic, sizeof kMagic) != 0) {
            munmap(m, (size_t)st.st_size);
            close(fd);
            return capi_fail(UCFP_E_INVALID, "%s is not a UCFP sidecar log", path);
        }
        const size_t good = walk(base, (size_t)st.st_size, [&](const Entry&) { entries++; });
        munmap(m, (size_t)st.st_size);
        if (good != (size_t)st.st_size && ftruncate(fd, (off_t)good) != 0) {
            close(fd);
            return capi_fail(UCFP_E_INDEX, "sidecar %s: cannot cut the torn tail: %s", path, strerror(errno));
        }
    }
    (void)lock_byte(fd, 1, F_UNLCK, false);           // the tail is final; byte 0 stays locked until close
    ucfp_sidecar* sc = new (std::nothrow) ucfp_sidecar();
    if (!sc) {
        close(fd);
        return capi_fail(UCFP_E_INDEX, "out of host memory");
    }
    sc->fd = fd;
    sc->entries = entries;
    *out = sc;
    return UCFP_OK;
}

void ucfp_sidecar_close(ucfp_sidecar* sc) {
    if (!sc) return;
    if (sc->fd >= 0) close(sc->fd);
    delete sc;
}

static int append(ucfp_sidecar* sc, uint8_t op, uint32_t tenant, uint64_t id, const uint8_t* fp, uint32_t fp_len, const float* emb,
                  uint32_t dim, const char* json, uint32_t json_len) {
    const uint64_t len64 = (uint64_t)kHead + fp_len + (uint64_t)dim * 4 + json_len;
    if (len64 > 0x7fffffffu) return capi_fail(UCFP_E_INVALID, "sidecar row of %llu bytes", (unsigned long long)len64);
    const uint32_t len = (uint32_t)len64;
    const size_t total = 8 + (((size_t)len + 7) & ~(size_t)7);
    std::lock_guard<std::mutex> lk(sc->mu);
    sc->buf.assign(total, 0);
    uint8_t* p = sc->buf.data() + 8;
    p[0] = op;
    memcpy(p + 4, &tenant, 4);
    memcpy(p + 8, &id, 8);
    memcpy(p + 16, &fp_len, 4);
    memcpy(p + 20, &dim, 4);
    memcpy(p + 24, &json_len, 4);
    if (fp_len) memcpy(p + kHead, fp, fp_len);
    if (dim) memcpy(p + kHead + fp_len, emb, (size_t)dim * 4);
    if (json_len) memcpy(p + kHead + fp_len + (size_t)dim * 4, json, json_len);
    const uint32_t crc = crc32_of(p, len);
    memcpy(sc->buf.data(), &len, 4);
    memcpy(sc->buf.data() + 4, &crc, 4);
    size_t done = 0;
    while (done < total) {                      // one entry = one write in the common case
        const ssize_t w = write(sc->fd, sc->buf.data() + done, total - done);
        if (w < 0) {
            if (errno == EINTR) continue;
            return capi_fail(UCFP_E_INDEX, "sidecar append: %s", strerror(errno));
        }
        done += (size_t)w;
    }
    sc->entries++;
    return UCFP_OK;
}

int ucfp_sidecar_append_upsert(ucfp_sidecar* sc, uint32_t tenant, uint64_t record_id, const uint8_t* fingerprint, uint32_t fp_len,
                               const float* embedding, uint32_t dim, const char* catalog_json, uint32_t json_len) {
    if (!sc || (fp_len && !fingerprint) || (dim && !embedding) || (json_len && !catalog_json))
        return capi_fail(UCFP_E_INVALID, "NULL argument");
    return append(sc, kUpsert, tenant, record_id, fingerprint, fp_len, embedding, dim, catalog_json, json_len);
}

int ucfp_sidecar_append_delete(ucfp_sidecar* sc, uint32_t tenant, uint64_t record_id) {
    if (!sc) return capi_fail(UCFP_E_INVALID, "sidecar is NULL");
    return append(sc, kDelete, tenant, record_id, nullptr, 0, nullptr, 0, nullptr, 0);
}

int ucfp_sidecar_sync(ucfp_sidecar* sc) {
    if (!sc) return capi_fail(UCFP_E_INVALID, "sidecar is NULL");
    std::lock_guard<std::mutex> lk(sc->mu);
    if (fdatasync(sc->fd) != 0) return capi_fail(UCFP_E_INDEX, "sidecar sync: %s", strerror(errno));
    return UCFP_OK;
}

int ucfp_sidecar_snapshot_open(const char* path, ucfp_sidecar_snapshot** out, uint64_t* live_rows, uint64_t* log_entries,
                               uint64_t* torn_bytes) {
    if (!path || !out) return capi_fail(UCFP_E_INVALID, "path/out is NULL");
    *out = nullptr;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return capi_fail(UCFP_E_NOT_FOUND, "sidecar %s: %s", path, strerror(errno));
    // no writer may cut the tail between our fstat and the end of the walk (a read lock needs no write access)
    struct TailLock {
        int fd;
        bool held;
        ~TailLock() { if (held) (void)lock_byte(fd, 1, F_UNLCK, false); }
    } tail{fd, lock_byte(fd, 1, F_RDLCK, true) == 0};
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof kMagic) {
        close(fd);
        return capi_fail(UCFP_E_INVALID, "%s is not a UCFP sidecar log", path);
    }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
        close(fd);
        return capi_fail(UCFP_E_INDEX, "sidecar %s: mmap: %s", path, strerror(errno));
    }
    const uint8_t* base = static_cast<const uint8_t*>(m);
    if (memcmp(base, kMagic, sizeof kMagic) != 0) {
        munmap(m, (size_t)st.st_size);
        close(fd);
        return capi_fail(UCFP_E_INVALID, "%s is not a UCFP sidecar log", path);
    }
    ucfp_sidecar_snapshot* s = new (std::nothrow) ucfp_sidecar_snapshot();
    if (!s) {
        munmap(m, (size_t)st.st_size);
        close(fd);
        return capi_fail(UCFP_E_INDEX, "out of host memory");
    }
    s->fd = fd;
    s->base = base;
    s->size = (size_t)st.st_size;
    std::unordered_map<std::pair<uint32_t, uint64_t>, Entry, KeyHash> live;
    uint64_t n = 0;
    const size_t good = walk(base, s->size, [&](const Entry& e) {
        n++;
        const auto key = std::make_pair(e.tenant, e.id);
        if (e.op == kUpsert) live[key] = e;
        else live.erase(key);
    });
    s->rows.reserve(live.size());
    for (auto& kv : live) s->rows.push_back(kv.second);
    std::sort(s->rows.begin(), s->rows.end(),
              [](const Entry& a, const Entry& b) { return a.tenant != b.tenant ? a.tenant < b.tenant : a.id < b.id; });
    if (live_rows) *live_rows = s->rows.size();
    if (log_entries) *log_entries = n;
    if (torn_bytes) *torn_bytes = s->size - good;
    *out = s;
    return UCFP_OK;
}

void ucfp_sidecar_snapshot_close(ucfp_sidecar_snapshot* s) {
    if (!s) return;
    if (s->base) munmap(const_cast<uint8_t*>(s->base), s->size);
    if (s->fd >= 0) close(s->fd);
    delete s;
}

int ucfp_sidecar_snapshot_row(ucfp_sidecar_snapshot* s, uint64_t i, uint32_t* tenant, uint64_t* record_id, const uint8_t** fingerprint,
                              uint32_t* fp_len, const uint8_t** embedding_bytes, uint32_t* dim, const char** catalog_json,
                              uint32_t* json_len) {
    if (!s) return capi_fail(UCFP_E_INVALID, "snapshot is NULL");
    if (i >= s->rows.size()) return capi_fail(UCFP_E_NOT_FOUND, "row %llu of %zu", (unsigned long long)i, s->rows.size());
    const Entry& e = s->rows[i];
    if (tenant) *tenant = e.tenant;
    if (record_id) *record_id = e.id;
    if (fingerprint) *fingerprint = e.fp;
    if (fp_len) *fp_len = e.fp_len;
    if (embedding_bytes) *embedding_bytes = reinterpret_cast<const uint8_t*>(e.emb);
    if (dim) *dim = e.dim;
    if (catalog_json) *catalog_json = e.json;
    if (json_len) *json_len = e.json_len;
    return UCFP_OK;
}

int ucfp_sidecar_snapshot_gather_fingerprints(ucfp_sidecar_snapshot* s, const char* algorithm, uint32_t fp_len, uint32_t* tenants,
                                              uint64_t* ids, uint8_t* fingerprints, uint64_t cap, uint64_t* n) {
    if (!s || !algorithm || !n) return capi_fail(UCFP_E_INVALID, "NULL argument");
    const size_t al = strlen(algorithm);
    uint64_t k = 0;
    for (const Entry& e : s->rows) {
        if (e.fp_len != fp_len) continue;
        const char* tag = nullptr;
        uint32_t tl = 0;
        if (!json_algorithm(e.json, e.json_len, &tag, &tl) || tl != al || memcmp(tag, algorithm, al) != 0) continue;
        if (k < cap) {
            if (tenants) tenants[k] = e.tenant;
            if (ids) ids[k] = e.id;
            if (fingerprints) memcpy(fingerprints + k * fp_len, e.fp, fp_len);
        }
        k++;
    }
    *n = k;        // rows that match; more than `cap`: call again with a larger buffer
    return UCFP_OK;
}

int ucfp_sidecar_snapshot_gather_vectors(ucfp_sidecar_snapshot* s, uint32_t dim, uint32_t* tenants, uint64_t* ids, float* rows,
                                         uint64_t cap, uint64_t* n) {
    if (!s || !n || dim == 0) return capi_fail(UCFP_E_INVALID, "NULL argument / dim 0");
    uint64_t k = 0;
    for (const Entry& e : s->rows) {
        if (e.dim != dim) continue;
        if (k < cap) {
            if (tenants) tenants[k] = e.tenant;
            if (ids) ids[k] = e.id;
            if (rows) memcpy(rows + k * dim, e.emb, (size_t)dim * 4);
        }
        k++;
    }
    *n = k;
    return UCFP_OK;
}

}  // extern "C"
