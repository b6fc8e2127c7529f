// Shared between the translation units of the host runtime (engine.h / engine.hip / engine_abi.hip: the codec; coder_abi.hip: the stand-alone coder and
// checkerboard operators of the C ABI).  Not part of the ABI.
#pragma once
#include <memory>
#include <shared_mutex>
#include <vector>

#include "common.h"

// One generation of device buffers: freed when the last engine instance that still points into it lets go.  A parent
// engine and its rgbd_elic_clone_shared() clones share generations, so re-uploading weights or tables on one of them
// (finalize / set_tables / set_scale_table build a NEW generation) can never free memory another one still reads.
struct DevGen {
    std::vector<void*> p;
    ~DevGen()
    {
        for (void* q : p) (void)hipFree(q);
    }
};

struct TableSet {
    DevTables d{};
    void* blob = nullptr;
    bool ready = false;
    int stride_src = 0;
    std::shared_ptr<DevGen> hold;  // owner of blob (engine table slots); rgbd_tables frees its blob itself
};

// packs the reference's CDF rows (entropy_models.py:196-204 via ops.cpp:24-81) with the coder's search / division tables
int build_tables(const int32_t* cdf, int stride, const int32_t* sizes, const int32_t* offsets, int nrows, TableSet* ts);

struct rgbd_tables {
    TableSet ts;
};

// Stream capture vs device-wide operations: a hipFree / hipDeviceSynchronize / synchronous hipMemcpy issued by ANY host
// thread while another thread's stream is capturing fails ("operation not permitted when stream is capturing") and
// poisons that capture.  Captures hold this lock shared (several engine instances may capture at once); everything that
// frees or synchronises device-wide takes it exclusively.
extern std::shared_mutex g_capture_mu;

int64_t rgbd_enc_cap_words(int64_t n);  // worst-case words of one stream of n symbols (rgbd_rans_max_bytes / 4)
