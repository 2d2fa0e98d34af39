/* lmdb_walk.h — the read-only walk of an LMDB data file the two shim examples share: for every arroy
 * ITEM key, the POINTER to the stored value inside its page — what `ImmutableLeafs::new` collects (src/parallel.rs:271-293).
 * Values sit at odd offsets, some on overflow pages.  Plain C99, header-only. */
#ifndef ARROY_HIP_EXAMPLES_LMDB_WALK_H
#define ARROY_HIP_EXAMPLES_LMDB_WALK_H
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>

/* ---- little helpers (the file and the host are little-endian; arroy's keys and child ids are big-endian) ------------- */
static uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

/* ---- 1. the LMDB walk ------------------------------------------------------------------------- */
typedef struct { uint32_t id; const uint8_t *ptr; size_t len; } item_rec;
typedef struct { const uint8_t *data; size_t size, psize; item_rec *items; size_t n, cap; } walker;

static int walk_page(walker *w, uint64_t pgno) {
    const uint8_t *pg = w->data + pgno * w->psize;
    if ((pgno + 1) * w->psize > w->size) return 1;
    const uint16_t flags = rd16(pg + 10), lower = rd16(pg + 12);
    const size_t n_nodes = (size_t)(lower - 16) / 2;
    for (size_t i = 0; i < n_nodes; i++) {
        const uint8_t *node = pg + rd16(pg + 16 + 2 * i);
        const uint16_t lo = rd16(node), hi = rd16(node + 2), nflags = rd16(node + 4), ksize = rd16(node + 6);
        const uint8_t *key = node + 8;
        if (flags & 0x01) { /* branch page: the child page number is spread over lo | hi | flags */
            if (walk_page(w, (uint64_t)lo | (uint64_t)hi << 16 | (uint64_t)nflags << 32)) return 1;
            continue;
        }
        const size_t dsize = (size_t)lo | (size_t)hi << 16;
        const uint8_t *val = key + ksize;
        if (nflags & 0x01) val = w->data + rd64(val) * w->psize + 16; /* F_BIGDATA: overflow page, payload 16 bytes in */
        /* arroy key (src/key.rs:56-71): [index u16 BE][mode u8][item u32 BE][padding]; mode 3 = Item */
        if (ksize >= 7 && key[2] == 3) {
            if (w->n == w->cap) {
                w->cap = w->cap ? 2 * w->cap : 128;
                w->items = (item_rec *)realloc(w->items, w->cap * sizeof(item_rec));
                if (!w->items) return 1;
            }
            w->items[w->n].id = be32(key + 3);
            w->items[w->n].ptr = val;
            w->items[w->n].len = dsize;
            w->n++;
        }
    }
    return 0;
}
static int by_id(const void *a, const void *b) {
    const uint32_t x = ((const item_rec *)a)->id, y = ((const item_rec *)b)->id;
    return x < y ? -1 : x > y;
}


/* Maps `path`, walks the main DB of the newer meta page, returns the item records in ascending id order (NULL on failure,
 * `why` says what).  The mapping stays for the life of the process, like an open `RoTxn`. */
static item_rec *lmdb_items(const char *path, size_t *n_out, size_t *psize_out, const char **why) {
    const int fd = open(path, O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) { *why = "cannot open the LMDB data file (run from the repository root or pass its path)"; return NULL; }
    const uint8_t *data = (const uint8_t *)mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (data == MAP_FAILED) { *why = "mmap failed"; return NULL; }
    /* meta pages 0 and 1: magic, version, address, mapsize, then the free and the main DB records, last page, txn id */
    if (rd32(data + 16) != 0xBEEFC0DEu) { *why = "not an LMDB data file"; return NULL; }
    const size_t psize = rd32(data + 16 + 24); /* mm_dbs[0].md_pad holds the page size */
    if (psize < 512 || (size_t)sb.st_size < 2 * psize) { *why = "implausible page size"; return NULL; }
    const uint64_t txn0 = rd64(data + 16 + 24 + 96 + 8), txn1 = rd64(data + psize + 16 + 24 + 96 + 8);
    const uint8_t *meta = txn1 > txn0 ? data + psize : data;
    const uint64_t root = rd64(meta + 16 + 24 + 48 + 40); /* main DB: md_root */
    walker w = {data, (size_t)sb.st_size, psize, NULL, 0, 0};
    if (walk_page(&w, root) != 0) { *why = "B-tree walk failed"; return NULL; }
    qsort(w.items, w.n, sizeof(item_rec), by_id);
    *n_out = w.n;
    *psize_out = psize;
    return w.items;
}
#endif
