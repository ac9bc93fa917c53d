// capi_strain.hip -- Strain / RegionFit engines and the reliable / unreliable selection of the RegionFit -> re-ICGN loop (part of the C-ABI of include/opencorr_hip.h; split from capi.hip in round 6, same exported symbols)
#include "capi_internal.h"

extern "C" {

// ---------------------------------------------------------------------------
// Strain
// ---------------------------------------------------------------------------
static int strain_check(const oc_hip_engine* e, float radius, int nmin, int approximation) {
    (void)e;
    if (!(radius > 0.f)) return fail(OC_HIP_ERR_INVALID, "Strain: subregion radius must be > 0 (got %g)", (double)radius);
    if (nmin < 1) return fail(OC_HIP_ERR_INVALID, "Strain: neighbor_number_min must be >= 1 (got %d)", nmin);
    if (nmin > ochip::strain_knn_max())
        return fail(OC_HIP_ERR_UNSUPPORTED, "Strain: neighbor_number_min %d exceeds the KNN path's limit of %d", nmin,
                    ochip::strain_knn_max());
    if (approximation != 1 && approximation != 2)
        return fail(OC_HIP_ERR_INVALID, "Strain: approximation must be 1 (Cauchy) or 2 (Green), got %d", approximation);
    return OC_HIP_OK;
}

int oc_hip_strain_create(float subregion_radius, int neighbor_number_min, int device, oc_hip_engine** out) {
    if (out) *out = nullptr;
    OC_TRY(strain_check(nullptr, subregion_radius, neighbor_number_min, 1));
    OC_TRY(create_engine(OC_HIP_STRAIN, 1, 1, 0, 0.f, 0.f, device, out));
    (*out)->st_radius = subregion_radius;
    (*out)->st_nmin = neighbor_number_min;
    return OC_HIP_OK;
}

int oc_hip_region_fit_create(float neighbor_search_radius, int neighbor_number_min, int device, oc_hip_engine** out) {
    if (out) *out = nullptr;
    OC_TRY(strain_check(nullptr, neighbor_search_radius, neighbor_number_min, 1));
    OC_TRY(create_engine(OC_HIP_REGION_FIT, 1, 1, 0, 0.f, 0.f, device, out));
    (*out)->st_radius = neighbor_search_radius;
    (*out)->st_nmin = neighbor_number_min;
    return OC_HIP_OK;
}

int oc_hip_region_fit_set(oc_hip_engine* e, float neighbor_search_radius, int neighbor_number_min) {
    OC_TRY(check_engine(e));
    if (e->kind != OC_HIP_REGION_FIT) return fail(OC_HIP_ERR_INVALID, "region_fit_set: not a RegionFit engine");
    OC_TRY(strain_check(e, neighbor_search_radius, neighbor_number_min, 1));
    std::lock_guard<std::mutex> lock(e->mu);
    if (neighbor_search_radius != e->st_radius) e->st_count = 0;
    e->st_radius = neighbor_search_radius;
    e->st_nmin = neighbor_number_min;
    return OC_HIP_OK;
}

int oc_hip_strain_set(oc_hip_engine* e, float subregion_radius, int neighbor_number_min, float zncc_threshold,
                      int approximation) {
    OC_TRY(check_engine(e));
    if (e->kind != OC_HIP_STRAIN) return fail(OC_HIP_ERR_INVALID, "strain_set: not a Strain engine");
    OC_TRY(strain_check(e, subregion_radius, neighbor_number_min, approximation));
    std::lock_guard<std::mutex> lock(e->mu);
    if (subregion_radius != e->st_radius) e->st_count = 0;  // the grid pitch follows the radius: prepare() again
    e->st_radius = subregion_radius;
    e->st_nmin = neighbor_number_min;
    e->st_zncc = zncc_threshold;
    e->st_approx = approximation;
    return OC_HIP_OK;
}

static int strain_stage(oc_hip_engine* e, const void* pois, size_t count, size_t stride_bytes, int ndim, int memory,
                        float** d_pois) {
    if (e->kind != OC_HIP_STRAIN && e->kind != OC_HIP_REGION_FIT) return fail(OC_HIP_ERR_INVALID, "not a Strain / RegionFit engine");
    if (ndim != 2 && ndim != 3) return fail(OC_HIP_ERR_INVALID, "ndim must be 2 (POI2D) or 3 (POI3D), got %d", ndim);
    if (!pois) return fail(OC_HIP_ERR_INVALID, "null POI buffer");
    const size_t rec = ndim == 2 ? OC_HIP_POI2D_BYTES : OC_HIP_POI3D_BYTES;
    if (stride_bytes < rec || (stride_bytes & 3))
        return fail(OC_HIP_ERR_INVALID, "bad POI stride %zu (record is %zu bytes, stride must be a multiple of 4)", stride_bytes, rec);
    if (count > 0x7fffffffull) return fail(OC_HIP_ERR_UNSUPPORTED, "Strain: at most 2^31-1 POIs per queue");
    if (memory == OC_HIP_DEVICE) {
        *d_pois = static_cast<float*>(const_cast<void*>(pois));
        return OC_HIP_OK;
    }
    OC_TRY(e->poi_stage.reserve(count * stride_bytes));
    OC_HIP_TRY(hipMemcpyAsync(e->poi_stage.p, pois, count * stride_bytes, hipMemcpyHostToDevice, e->stream));
    *d_pois = e->poi_stage.as<float>();
    return OC_HIP_OK;
}

// neighbour search over a queue's coordinates; gather_records: also snapshot the fit records (RegionFit's cloud)
static int plane_prepare(oc_hip_engine* e, int kind, const void* pois, size_t count, size_t stride_bytes, int ndim, int memory,
                         bool gather_records) {
    OC_ACTIVATE(e);
    if (e->kind != kind) return fail(OC_HIP_ERR_INVALID, "prepare: wrong engine kind for this entry point");
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    e->st_count = 0;
    if (count == 0) return OC_HIP_OK;
    OC_TRY(order_after_default_stream(e));
    float* d_pois = nullptr;
    OC_TRY(strain_stage(e, pois, count, stride_bytes, ndim, memory, &d_pois));
    const int stride_f = (int)(stride_bytes / 4);
    // bounding box -> grid (the cell count is needed on the host to size the tables)
    OC_TRY(e->st_box.reserve(6 * sizeof(unsigned)));
    OC_HIP_TRY(ochip::launch_strain_bbox(ndim, d_pois, stride_f, count, e->st_box.as<unsigned>(), e->stream));
    unsigned box[6];
    OC_HIP_TRY(hipMemcpyAsync(box, e->st_box.p, sizeof(box), hipMemcpyDeviceToHost, e->stream));
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    const ochip::StrainGrid g = ochip::strain_make_grid(ndim, box, e->st_radius);
    const size_t ncell = ochip::strain_cell_count(g);
    OC_TRY(e->st_counts.reserve(ncell * sizeof(unsigned)));
    OC_TRY(e->st_cursor.reserve(ncell * sizeof(unsigned)));
    OC_TRY(e->st_start.reserve((ncell + 1) * sizeof(unsigned)));
    OC_TRY(e->st_slots.reserve(count * sizeof(unsigned)));
    OC_TRY(e->st_order.reserve(count * sizeof(unsigned)));
    OC_HIP_TRY(ochip::launch_strain_sort(ndim, d_pois, stride_f, count, g, e->st_counts.as<unsigned>(), e->st_start.as<unsigned>(),
                                         e->st_cursor.as<unsigned>(), e->st_slots.as<unsigned>(), e->st_order.as<unsigned>(),
                                         e->stream));
    if (gather_records) {
        OC_TRY(e->st_recs.reserve(count * 32));
        OC_HIP_TRY(ochip::launch_strain_gather(ndim, d_pois, stride_f, count, e->st_order.as<unsigned>(), e->st_recs.p, e->stream));
    }
    if (memory == OC_HIP_HOST) OC_HIP_TRY(hipStreamSynchronize(e->stream));  // the staging buffer is reused by compute
    else OC_TRY(finish_device_call(e));
    e->st_grid = g;
    e->st_ndim = ndim;
    e->st_count = count;
    return OC_HIP_OK;
}

int oc_hip_strain_prepare(oc_hip_engine* e, const void* pois, size_t count, size_t stride_bytes, int ndim, int memory) {
    OC_TRY(check_engine(e));
    return plane_prepare(e, OC_HIP_STRAIN, pois, count, stride_bytes, ndim, memory, false);
}

int oc_hip_region_fit_prepare(oc_hip_engine* e, const void* reliable_pois, size_t count, size_t stride_bytes, int ndim,
                              int memory) {
    OC_TRY(check_engine(e));
    return plane_prepare(e, OC_HIP_REGION_FIT, reliable_pois, count, stride_bytes, ndim, memory, true);
}

int oc_hip_region_fit_compute(oc_hip_engine* e, void* pois, size_t count, size_t stride_bytes, int ndim, int memory) {
    OC_ACTIVATE(e);
    if (count == 0) return OC_HIP_OK;
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    if (e->kind != OC_HIP_REGION_FIT) return fail(OC_HIP_ERR_INVALID, "not a RegionFit engine");
    if (e->st_count == 0)
        return fail(OC_HIP_ERR_INVALID, "RegionFit: setNeighbor + prepare has not been called (or the radius changed since)");
    if (e->st_ndim != ndim) return fail(OC_HIP_ERR_INVALID, "RegionFit: prepared for POI%dD, compute() got POI%dD", e->st_ndim, ndim);
    OC_TRY(order_after_default_stream(e));
    float* d_pois = nullptr;
    OC_TRY(strain_stage(e, pois, count, stride_bytes, ndim, memory, &d_pois));
    const int stride_f = (int)(stride_bytes / 4);
    OC_TRY(e->st_fallback.reserve((count + 1) * sizeof(unsigned)));
    const ochip::StrainParams P = {e->st_radius * e->st_radius, 0.f, e->st_nmin, 1};
    {
        ProfScope prof(e);
        OC_HIP_TRY(ochip::launch_region_fit_compute(ndim, d_pois, stride_f, count, e->st_grid, P, e->st_start.as<unsigned>(),
                                                    e->st_recs.p, e->st_fallback.as<unsigned>(), e->stream));
    }
    if (memory == OC_HIP_HOST) {
        OC_HIP_TRY(hipMemcpyAsync(pois, d_pois, count * stride_bytes, hipMemcpyDeviceToHost, e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
    } else {
        OC_TRY(finish_device_call(e));
    }
    return OC_HIP_OK;
}

int oc_hip_strain_compute(oc_hip_engine* e, void* pois, size_t count, size_t stride_bytes, int ndim, int memory) {
    OC_ACTIVATE(e);
    if (count == 0) return OC_HIP_OK;
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    if (e->kind != OC_HIP_STRAIN) return fail(OC_HIP_ERR_INVALID, "not a Strain engine");
    if (e->st_count == 0) return fail(OC_HIP_ERR_INVALID, "Strain: prepare(poi_queue) has not been called (or the radius changed since)");
    if (e->st_count != count || e->st_ndim != ndim)
        return fail(OC_HIP_ERR_INVALID, "Strain: prepare() saw %zu POI%dD, compute() got %zu POI%dD", e->st_count, e->st_ndim, count, ndim);
    OC_TRY(order_after_default_stream(e));
    float* d_pois = nullptr;
    OC_TRY(strain_stage(e, pois, count, stride_bytes, ndim, memory, &d_pois));
    const int stride_f = (int)(stride_bytes / 4);
    OC_TRY(e->st_recs.reserve(count * 32));
    OC_TRY(e->st_fallback.reserve((count + 1) * sizeof(unsigned)));
    const ochip::StrainParams P = {e->st_radius * e->st_radius, e->st_zncc, e->st_nmin, e->st_approx};
    {
        ProfScope prof(e);
        OC_HIP_TRY(ochip::launch_strain_compute(ndim, d_pois, stride_f, count, e->st_grid, P, e->st_start.as<unsigned>(),
                                                e->st_order.as<unsigned>(), e->st_recs.p, e->st_fallback.as<unsigned>(), e->stream));
    }
    if (memory == OC_HIP_HOST) {
        OC_HIP_TRY(hipMemcpyAsync(pois, d_pois, count * stride_bytes, hipMemcpyDeviceToHost, e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
    } else {
        OC_TRY(finish_device_call(e));
    }
    return OC_HIP_OK;
}


// ---------------------------------------------------------------------------
// reliable / unreliable selection of the RegionFit -> re-ICGN loop (poi_split.hip)
// ---------------------------------------------------------------------------
static int split_params(int ndim, size_t stride_bytes, float low, float high, float conv, int mode, ochip::PoiSplitParams* P) {
    if (ndim != 2 && ndim != 3) return fail(OC_HIP_ERR_INVALID, "ndim must be 2 (POI2D) or 3 (POI3D), got %d", ndim);
    const size_t rec = ndim == 2 ? OC_HIP_POI2D_BYTES : OC_HIP_POI3D_BYTES;
    if (stride_bytes < rec || (stride_bytes & 3)) return fail(OC_HIP_ERR_INVALID, "bad POI stride %zu (record is %zu bytes, stride must be a multiple of 4)", stride_bytes, rec);
    P->mode = mode;
    P->rec_floats = (int)(rec / 4);
    P->zncc_at = ndim == 2 ? 16 : 18;  // result.zncc / result.convergence, src/oc_poi.h:102-136, 187-222
    P->conv_at = ndim == 2 ? 18 : 20;
    P->zncc_low = low;
    P->zncc_high = high;
    P->conv = conv;
    return OC_HIP_OK;
}

// totals[0], totals[1]: records of class 0 / 1; totals[2]: a main-queue index was out of range (merge_recovered)
static int read_split_totals(oc_hip_engine* e, size_t count, size_t totals[3]) {
    unsigned host[3] = {0, 0, 0};
    const unsigned* d = e->split_scratch.as<unsigned>() + ochip::poi_split_scratch_words(count) - 3;
    OC_HIP_TRY(hipMemcpyAsync(host, d, sizeof(host), hipMemcpyDeviceToHost, e->stream));
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    totals[0] = host[0];
    totals[1] = host[1];
    totals[2] = host[2];
    return OC_HIP_OK;
}

// device records (stride_bytes apart) -> the caller's host queue: only the record's own bytes travel, so whatever the
// caller keeps between records (stride_bytes > record size) stays as it was -- like on the DEVICE path, whose scatter
// writes rec_floats per record
static hipError_t copy_records_to_host(void* dst, const void* src, size_t n, size_t stride_bytes, size_t rec_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (stride_bytes == rec_bytes) return hipMemcpyAsync(dst, src, n * stride_bytes, hipMemcpyDeviceToHost, stream);
    return hipMemcpy2DAsync(dst, stride_bytes, src, stride_bytes, rec_bytes, n, hipMemcpyDeviceToHost, stream);
}

int oc_hip_split_reliable(oc_hip_engine* e, const void* pois, size_t count, size_t stride_bytes, int ndim, float zncc_threshold_low,
                          float zncc_threshold_high, float conv_criterion, void* reliable, size_t reliable_offset, void* unreliable,
                          unsigned* unreliable_index, size_t* n_reliable, size_t* n_unreliable, int memory) {
    OC_ACTIVATE(e);
    if (!n_reliable || !n_unreliable) return fail(OC_HIP_ERR_INVALID, "split_reliable: null count pointer");
    *n_reliable = *n_unreliable = 0;
    if (count == 0) return OC_HIP_OK;
    if (!pois || !reliable || !unreliable || !unreliable_index) return fail(OC_HIP_ERR_INVALID, "split_reliable: null buffer");
    ochip::PoiSplitParams P;
    OC_TRY(split_params(ndim, stride_bytes, zncc_threshold_low, zncc_threshold_high, conv_criterion, 0, &P));
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));
    OC_TRY(e->split_scratch.reserve(ochip::poi_split_scratch_words(count) * sizeof(unsigned)));
    const int stride_f = (int)(stride_bytes / 4);
    const size_t rec_bytes = (size_t)P.rec_floats * 4;
    size_t totals[3];
    if (memory == OC_HIP_DEVICE) {
        OC_HIP_TRY(ochip::launch_poi_split(static_cast<const float*>(pois), stride_f, count, P, nullptr, static_cast<float*>(reliable),
                                           reliable_offset, nullptr, static_cast<float*>(unreliable), unreliable_index, nullptr, 0,
                                           e->split_scratch.as<unsigned>(), e->stream));
        OC_TRY(read_split_totals(e, count, totals));
    } else {
        const size_t qb = count * stride_bytes;
        OC_TRY(e->poi_stage.reserve(3 * qb + count * sizeof(unsigned)));
        char* base = e->poi_stage.as<char>();
        float* d_in = reinterpret_cast<float*>(base);
        float* d_rel = reinterpret_cast<float*>(base + qb);
        float* d_unr = reinterpret_cast<float*>(base + 2 * qb);
        unsigned* d_idx = reinterpret_cast<unsigned*>(base + 3 * qb);
        OC_HIP_TRY(hipMemcpyAsync(d_in, pois, qb, hipMemcpyHostToDevice, e->stream));
        OC_HIP_TRY(ochip::launch_poi_split(d_in, stride_f, count, P, nullptr, d_rel, 0, nullptr, d_unr, d_idx, nullptr, 0,
                                           e->split_scratch.as<unsigned>(), e->stream));
        OC_TRY(read_split_totals(e, count, totals));
        OC_HIP_TRY(copy_records_to_host(static_cast<char*>(reliable) + reliable_offset * stride_bytes, d_rel, totals[0], stride_bytes, rec_bytes, e->stream));
        if (totals[1]) {
            OC_HIP_TRY(copy_records_to_host(unreliable, d_unr, totals[1], stride_bytes, rec_bytes, e->stream));
            OC_HIP_TRY(hipMemcpyAsync(unreliable_index, d_idx, totals[1] * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
        }
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
    }
    *n_reliable = totals[0];
    *n_unreliable = totals[1];
    return OC_HIP_OK;
}

int oc_hip_merge_recovered(oc_hip_engine* e, void* pois, size_t count, size_t stride_bytes, int ndim, void* unreliable,
                           unsigned* unreliable_index, size_t n_unreliable, float zncc_threshold_high, float conv_criterion, void* reliable,
                           size_t reliable_offset, size_t* n_recovered, size_t* n_remaining, int memory) {
    OC_ACTIVATE(e);
    if (!n_recovered || !n_remaining) return fail(OC_HIP_ERR_INVALID, "merge_recovered: null count pointer");
    *n_recovered = 0;
    *n_remaining = 0;
    if (n_unreliable == 0) return OC_HIP_OK;
    if (!pois || !reliable || !unreliable || !unreliable_index) return fail(OC_HIP_ERR_INVALID, "merge_recovered: null buffer");
    ochip::PoiSplitParams P;
    OC_TRY(split_params(ndim, stride_bytes, 0.f, zncc_threshold_high, conv_criterion, 1, &P));
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));
    OC_TRY(e->split_scratch.reserve(ochip::poi_split_scratch_words(n_unreliable) * sizeof(unsigned)));
    const int stride_f = (int)(stride_bytes / 4);
    const size_t qb = n_unreliable * stride_bytes, ib = n_unreliable * sizeof(unsigned);
    const size_t rec_bytes = (size_t)P.rec_floats * 4;
    size_t totals[3];
    if (memory == OC_HIP_DEVICE) {
        // the POIs that stay unreliable are compacted into a scratch copy first (an in-place compaction would overwrite
        // records other threads still have to read), then moved back to the front of the caller's arrays
        OC_TRY(e->split_tmp.reserve(qb + ib));
        float* t_rec = e->split_tmp.as<float>();
        unsigned* t_idx = reinterpret_cast<unsigned*>(e->split_tmp.as<char>() + qb);
        // the index list is checked on the device BEFORE the scatter may write anything of the caller's: a list
        // with an entry outside the main queue is refused with `reliable`, `pois`, `unreliable` and the list itself untouched
        {
            unsigned* flag = e->split_scratch.as<unsigned>() + ochip::poi_split_scratch_words(n_unreliable) - 1;
            OC_HIP_TRY(ochip::launch_poi_index_range(unreliable_index, n_unreliable, count, flag, e->stream));
            unsigned bad = 0;
            OC_HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(bad), hipMemcpyDeviceToHost, e->stream));
            OC_HIP_TRY(hipStreamSynchronize(e->stream));
            if (bad) return fail(OC_HIP_ERR_INVALID, "merge_recovered: an unreliable_index entry is >= the main queue's %zu records (nothing was changed)", count);
        }
        OC_HIP_TRY(ochip::launch_poi_split(static_cast<const float*>(unreliable), stride_f, n_unreliable, P, unreliable_index,
                                           static_cast<float*>(reliable), reliable_offset, nullptr, t_rec, t_idx, static_cast<float*>(pois),
                                           count, e->split_scratch.as<unsigned>(), e->stream));
        OC_TRY(read_split_totals(e, n_unreliable, totals));
        // an index outside the main queue: the kernel wrote nothing through it; the caller's lists are left as they were
        if (totals[2]) return fail(OC_HIP_ERR_INVALID, "merge_recovered: an unreliable_index entry is >= the main queue's %zu records", count);
        if (totals[1]) {
            OC_HIP_TRY(hipMemcpyAsync(unreliable, t_rec, totals[1] * stride_bytes, hipMemcpyDeviceToDevice, e->stream));
            OC_HIP_TRY(hipMemcpyAsync(unreliable_index, t_idx, totals[1] * sizeof(unsigned), hipMemcpyDeviceToDevice, e->stream));
        }
        *n_recovered = totals[0];
        *n_remaining = totals[1];
        return finish_device_call(e);
    }
    // host queues: every index is checked BEFORE anything is enqueued or touched (the host knows the list)
    for (size_t j = 0; j < n_unreliable; j++)
        if (unreliable_index[j] >= count)
            return fail(OC_HIP_ERR_INVALID, "merge_recovered: unreliable_index[%zu] = %u is >= the main queue's %zu records", j, unreliable_index[j], count);
    // the classification and both compactions run on the device; the host only moves the recovered records to where the
    // device's index list says they go
    OC_TRY(e->poi_stage.reserve(3 * qb + 3 * ib));
    char* base = e->poi_stage.as<char>();
    float* d_in = reinterpret_cast<float*>(base);
    float* d_rec = reinterpret_cast<float*>(base + qb);
    float* d_rem = reinterpret_cast<float*>(base + 2 * qb);
    unsigned* d_idx_in = reinterpret_cast<unsigned*>(base + 3 * qb);
    unsigned* d_idx_rec = d_idx_in + n_unreliable;
    unsigned* d_idx_rem = d_idx_rec + n_unreliable;
    OC_HIP_TRY(hipMemcpyAsync(d_in, unreliable, qb, hipMemcpyHostToDevice, e->stream));
    OC_HIP_TRY(hipMemcpyAsync(d_idx_in, unreliable_index, ib, hipMemcpyHostToDevice, e->stream));
    OC_HIP_TRY(ochip::launch_poi_split(d_in, stride_f, n_unreliable, P, d_idx_in, d_rec, 0, d_idx_rec, d_rem, d_idx_rem, nullptr, 0,
                                       e->split_scratch.as<unsigned>(), e->stream));
    OC_TRY(read_split_totals(e, n_unreliable, totals));
    std::vector<unsigned> rec_idx(totals[0]);
    char* rel_dst = static_cast<char*>(reliable) + reliable_offset * stride_bytes;
    if (totals[0]) {
        OC_HIP_TRY(copy_records_to_host(rel_dst, d_rec, totals[0], stride_bytes, rec_bytes, e->stream));
        OC_HIP_TRY(hipMemcpyAsync(rec_idx.data(), d_idx_rec, totals[0] * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    }
    if (totals[1]) {
        OC_HIP_TRY(copy_records_to_host(unreliable, d_rem, totals[1], stride_bytes, rec_bytes, e->stream));
        OC_HIP_TRY(hipMemcpyAsync(unreliable_index, d_idx_rem, totals[1] * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    }
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    for (size_t j = 0; j < totals[0]; j++)
        std::memcpy(static_cast<char*>(pois) + (size_t)rec_idx[j] * stride_bytes, rel_dst + j * stride_bytes, rec_bytes);
    *n_recovered = totals[0];
    *n_remaining = totals[1];
    return OC_HIP_OK;
}


}  // extern "C"
