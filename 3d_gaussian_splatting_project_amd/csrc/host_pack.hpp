// Host side of the segmentation-map hand-over (gsx_vote_view with a HOST pointer): a small worker pool and the
// dtype-narrowing copy into the pinned staging ring.  Pure C++ (no HIP): compiled by g++.
//
// A map must cross PCIe; before that it must be copied out of the caller's pageable buffer into pinned memory
// anyway.  That one pass over the int32 / int64 labels (dls.py:124,142,158) writes the library's on-device form
// directly — u8 bins (label + 1) in strips of 16 pixel columns plus the 4x4-coarsened level — so that 2.2 MB
// instead of 8.3 MB per 1080p map go over the link and the GPU has nothing left to do but the DMA.  This is data
// marshalling for the boundary, not a CPU path of the labeler: projection, votes and arg-max exist in HIP only.
#pragma once
#include <cstddef>
#include <cstdint>

namespace gsx {

// Geometry of one packed map inside the seg pool (shared by the host packer, the device packer and vote_view).
struct MapLayout {
    int w = 0, h = 0;
    int strip_bytes = 0;    // bytes per strip of 16 pixel columns: rows padded to a multiple of 8, 16 B per row; 0 = row-major
    int cstrip_bytes = 0;   // same for the coarse level (strips of 16 cells = 64 pixel columns); 0 = no coarse level
    int cw = 0, ch = 0;     // coarse cells per row / column
    size_t fine_bytes = 0;  // bytes of the full-resolution level
    size_t coarse_off = 0;  // offset of the coarse level from the map's start (256-B aligned)
    size_t map_bytes = 0;   // total
};
MapLayout map_layout(int w, int h, bool tiled, bool coarse);

// Fork-join pool: run(n, fn, arg) calls fn(arg, part) for part in [0, n) on the pool's threads and the caller.
class Workers {
public:
    // threads: total parallelism including the calling thread (>= 1); numa_node: where the data to work on lives
    // (< 0: wherever the creating thread runs)
    explicit Workers(int threads, int numa_node = -1);
    ~Workers();
    Workers(const Workers&) = delete;
    Workers& operator=(const Workers&) = delete;
    int threads() const { return nthreads_; }
    void run(int parts, void (*fn)(void*, int), void* arg);

private:
    struct Impl;
    Impl* impl_;
    int nthreads_;
};
int numa_node_of(const void* addr);  // NUMA node holding that page, -1 if unknown
int default_host_threads();  // min(16, (CPUs this process may run on, capped by the cgroup quota) / LOCAL_WORLD_SIZE), or GSX_HOST_THREADS

// seg dtype codes as in gsx.h: 0 = int32, 1 = int64, 2 = u8 holding label+1, 3 = u8 holding the label.
// Writes the packed map (layout L) at dst; returns 0, or 1 if a label lies outside [-1, bins-2].
// `pool` may be nullptr (single thread).
int host_pack_map(Workers* pool, const void* seg, int seg_dtype, const MapLayout& L, int bins, uint8_t* dst);
// The COMPACT transfer form of a two-level map (what crosses PCIe for a host map; tiled + coarse layouts only):
//   [0, table_bytes)                     uint32 first_block[2 * bands]: where the blocks of each cell row start in the
//                                        stream (a band of 8 pixel rows holds two cell rows: entry 2 * band + row)
//   [table_bytes, +coarse_bytes)         the coarse level exactly as it sits in the pool (padding zeroed)
//   [stream_off, stream_off + 16*blocks) one 16-byte block (4 rows x 4 pixels, u8 bins) per MIXED 4x4 cell (coarse byte 255:
//                                        the cell's pixels differ, or it sticks out of the map); inside a cell row by cell
//                                        column, the cell rows in the order their workers reserved room (an atomic counter:
//                                        the order differs from run to run, the table says where)
// Uniform cells travel as their coarse byte alone; the GPU rebuilds the full-resolution level (seg_expand_kernel).
struct CompactLayout {
    int bands = 0;             // bands of 8 pixel rows
    size_t table_bytes = 0;    // 256-B aligned pieces
    size_t coarse_bytes = 0;
    size_t stream_off = 0;
    size_t capacity = 0;       // stream_off + room for every cell being mixed
};
CompactLayout compact_layout(const MapLayout& L);
void set_host_prefetch_burst(bool on);  // process-wide A/B switch: a band that starts a new stream asks for a whole prefetch distance at once (default on)
void set_host_prefetch(int bytes);       // process-wide: how far ahead the narrowing loops prefetch the map (default 8 KB)  // process-wide: parts per map the cut aims for (default 256; 1 = one part per band)
// Packs `seg` into the compact form at rec (capacity bytes), using `scratch` (L.fine_bytes bytes of ordinary memory: the
// narrowed strips live there for the duration of the call).  *blocks = 16-byte blocks in the stream.
// Returns 0, or 1 if a label lies outside [-1, bins-2].
int host_pack_map_compact(Workers* pool, const void* seg, int seg_dtype, const MapLayout& L, int bins, uint8_t* scratch,
                          uint8_t* rec, size_t* blocks);
// out = in, split over the pool (D2H epilogue: pinned -> the caller's pageable array)
void host_copy(Workers* pool, void* dst, const void* src, size_t bytes);
// dst[i] = (int)src[i] - 1 (bin -> label), split over the pool (D2H epilogue: the labels cross PCIe as one byte each)
void host_widen_labels(Workers* pool, int32_t* dst, const uint8_t* src, size_t n);
// test hook: `runs` fork-joins of varying size on ONE pool; returns the number of parts that did not run exactly once
long long workers_stress(int threads, int runs, int max_parts);

}  // namespace gsx
