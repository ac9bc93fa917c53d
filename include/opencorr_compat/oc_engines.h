// oc_engines.h -- OpenCorr's hot-path classes as thin shims over the HIP C-ABI.
//
// Same class names, constructor signatures and call order as the reference:
//   DIC / DVC bases                     src/oc_dic.h:43-84
//   FFTCC2D(int,int,int), FFTCC3D(int,int,int,int)                     src/oc_fftcc.h:54-89
//   ICGN2D1/2D2(int,int,float,float,int), ICGN3D1(int,int,int,float,float,int)   src/oc_icgn.h:45-180
// plus prepare(), prepareRef(), prepareTar(), compute(POI*), compute(std::vector<POI>&),
// setIteration(float,float), setIteration(POI*), setImages, setSubset.
// `thread_number` is kept for signature compatibility (the GPU parallelises internally);
// the device is chosen with the extra setters setDevice() / setDevices() or the OC_HIP_DEVICE / OC_HIP_DEVICES
// environment variables (OC_HIP_DEVICES=all: every engine spreads its queues over all GPUs of the node; OC_HIP_ARITH_FMA=1: the
// ICGN / IC-LM solvers run under the fused arithmetic contract, see oc_hip_set_tuning "arith_fma").  Failures of the engine (no GPU, bad call order, ...)
// are thrown as std::string like the reference does (src/oc_fftcc.cpp:145, src/oc_icgn.cpp:65).
//
// Differences a caller can observe (documented):
//   * images are snapshotted to HBM when needed (first prepare()/compute() after setImages);
//     editing the host image afterwards requires setImages() again, as with the reference's CUDA
//     module (examples/test_2d_dic_gpu_icgn.cpp:99-136);
//     engines that were given the SAME Image2D / Image3D pair share one device copy of it (the second engine's upload is
//     skipped when the first one's snapshot was taken after the second engine's setImages() call -- a read of the host
//     images "at some point after setImages()", which is all the contract above promises);
//   * setSelfAdaptive(true) is honoured by ICGN2D1/ICGN2D2; a POI whose radius is negative is rejected
//     with zncc = -3 (the reference would try to allocate a negative-sized subset).
#pragma once

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../opencorr_hip.h"
#include "oc_types.h"

namespace opencorr {

namespace hipdetail {
inline void check(int status) {
    if (status != OC_HIP_OK) throw std::string(oc_hip_last_error());
}
inline int default_device() {
    const char* e = std::getenv("OC_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}
// OC_HIP_DEVICES = "0,1,2,3" or "all": every engine created through these classes becomes a device group over the
// listed GPUs (oc_hip_set_devices) -- an unmodified OpenCorr main then uses the whole node.
inline std::vector<int> default_devices() {
    std::vector<int> ids;
    const char* e = std::getenv("OC_HIP_DEVICES");
    if (!e || !*e) return ids;
    if (std::string(e) == "all") {
        int n = 0;
        check(oc_hip_device_count(&n));
        for (int i = 0; i < n; i++) ids.push_back(i);
        return ids;
    }
    for (const char* p = e; *p;) {
        char* end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end == p) break;
        ids.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
    }
    return ids;
}
// Device snapshots of host image pairs, so that an FFTCC and an ICGN engine working on one pair upload it once (config B:
// 2 x 67 MB instead of 4 x 67 MB over PCIe; the reference shares the images the same way -- both engines hold pointers to
// the caller's Image2D, src/oc_dic.cpp:22-26).  An engine that needs its images looks for a snapshot of the same pair
// (same host pointers, same shape, same device) taken AFTER its own setImages() call (`seq` is a process-wide counter
// stamped on both events) and borrows it with oc_hip_share_images -- the device copy is reference counted inside the
// library, so the donor may be destroyed first.
struct Snapshot {
    const void* ref;
    const void* tar;
    int dims[3];
    int device;
    oc_hip_engine* donor;
    unsigned long long seq;
};
struct SnapshotRegistry {
    std::mutex mu;
    std::vector<Snapshot> entries;
    unsigned long long seq = 0;
    static SnapshotRegistry& get() {
        static SnapshotRegistry r;
        return r;
    }
    unsigned long long stamp() {
        std::lock_guard<std::mutex> lock(mu);
        return ++seq;
    }
    // a snapshot of this pair newer than `after`, borrowed into `engine`; false: the caller uploads itself
    bool borrow(oc_hip_engine* engine, const void* ref, const void* tar, int d0, int d1, int d2, int device, unsigned long long after) {
        std::lock_guard<std::mutex> lock(mu);
        for (const Snapshot& s : entries)
            if (s.ref == ref && s.tar == tar && s.dims[0] == d0 && s.dims[1] == d1 && s.dims[2] == d2 && s.device == device &&
                s.donor != engine && s.seq > after)
                return oc_hip_share_images(engine, s.donor) == OC_HIP_OK;
        return false;
    }
    void publish(oc_hip_engine* engine, const void* ref, const void* tar, int d0, int d1, int d2, int device) {
        std::lock_guard<std::mutex> lock(mu);
        forget_locked(engine);
        entries.push_back(Snapshot{ref, tar, {d0, d1, d2}, device, engine, ++seq});
    }
    void forget(oc_hip_engine* engine) {
        std::lock_guard<std::mutex> lock(mu);
        forget_locked(engine);
    }

private:
    void forget_locked(oc_hip_engine* engine) {
        for (size_t i = 0; i < entries.size();)
            if (entries[i].donor == engine) entries.erase(entries.begin() + (long)i);
            else i++;
    }
};
inline bool single_device(oc_hip_engine* e) {
    int n = 1;
    return oc_hip_get_devices(e, nullptr, 0, &n) == OC_HIP_OK && n == 1;
}
// applied by every shim constructor right after the engine exists
// true when OC_HIP_ARITH_FMA switches the solvers created through these classes to the fused arithmetic contract
inline bool arithFmaFromEnvironment() {
    const char* fma = std::getenv("OC_HIP_ARITH_FMA");
    return fma && std::atoi(fma) != 0;
}

inline void apply_default_devices(oc_hip_engine* e) {
    const std::vector<int> ids = default_devices();
    if (ids.size() > 1) check(oc_hip_set_devices(e, ids.data(), (int)ids.size()));
    // OC_HIP_ARITH_FMA=1: the ICGN / IC-LM solvers created through these classes use the fused arithmetic contract
    // (oc_hip_set_tuning "arith_fma", include/opencorr_hip.h) -- an unmodified OpenCorr main opts in without a source change;
    // engines without a fused build (FFTCC, NR2D1, Strain, RegionFit) are left alone
    const char* fma = std::getenv("OC_HIP_ARITH_FMA");
    if (fma && std::atoi(fma) != 0) {
        int kind = 0;
        if (oc_hip_get_kind(e, &kind) == OC_HIP_OK &&
            (kind == OC_HIP_ICGN2D1 || kind == OC_HIP_ICGN2D2 || kind == OC_HIP_ICLM2D1 || kind == OC_HIP_ICLM2D2 || kind == OC_HIP_ICGN3D1)) {
            check(oc_hip_set_tuning(e, "arith_fma", 1));
            // the one knob that changes result bits, set from outside the program: say so once per process;
            // arithFmaFromEnvironment() tells a caller which mode its engines run in
            static std::atomic<bool> said{false};
            const char* quiet = std::getenv("OC_HIP_QUIET");
            if (!(quiet && *quiet && *quiet != '0') && !said.exchange(true))
                std::fprintf(stderr, "opencorr_hip: OC_HIP_ARITH_FMA=%s -- the ICGN / IC-LM solvers of this process use the fused arithmetic contract "
                                     "(per-sample multiply-adds fused; results differ from the default build by rounding, inside 1e-4 px).  "
                                     "OC_HIP_QUIET=1 silences this note.\n", fma);
        }
    }
}
}  // namespace hipdetail

class DIC {
public:
    Image2D* ref_img = nullptr;
    Image2D* tar_img = nullptr;
    int subset_radius_x = 0, subset_radius_y = 0;
    int thread_number = 1;
    bool self_adaptive = false;

    DIC() {}
    virtual ~DIC() {
        if (engine_) {
            hipdetail::SnapshotRegistry::get().forget(engine_);
            oc_hip_destroy(engine_);
        }
    }
    DIC(const DIC&) = delete;
    DIC& operator=(const DIC&) = delete;

    void setImages(Image2D& ref, Image2D& tar) {
        ref_img = &ref;
        tar_img = &tar;
        images_dirty_ = true;
        images_seq_ = hipdetail::SnapshotRegistry::get().stamp();
        if (engine_) hipdetail::SnapshotRegistry::get().forget(engine_);  // this engine's snapshot is of older content
    }
    void setSubset(int radius_x, int radius_y) {
        subset_radius_x = radius_x;
        subset_radius_y = radius_y;
        if (engine_) hipdetail::check(oc_hip_set_subset(engine_, radius_x, radius_y, 0));
    }
    // DIC::setSelfAdaptive (src/oc_dic.cpp:34-37).  Honoured by ICGN2D1/ICGN2D2 (per-POI radius from
    // poi->subset_radius); FFTCC2D ignores it, as in the reference (src/oc_fftcc.cpp:177-275 never reads it).
    virtual void setSelfAdaptive(bool is_self_adaptive) { self_adaptive = is_self_adaptive; }
    // Moves the engine to another GPU / spreads it over several GPUs of the node (contiguous blocks of every queue,
    // src/oc_icgn.cpp:343-351 is the loop being shared out).  Images are uploaded again on the next use.
    void setDevice(int device) { setDevices(std::vector<int>(1, device)); }
    void setDevices(const std::vector<int>& devices) {
        if (devices.empty()) throw std::string("setDevices: empty device list");
        hipdetail::SnapshotRegistry::get().forget(engine_);
        hipdetail::check(oc_hip_set_devices(engine_, devices.data(), (int)devices.size()));
        device_ = devices[0];
        images_dirty_ = ref_img != nullptr;
    }

    virtual void prepare() = 0;
    virtual void compute(POI2D* poi) = 0;
    virtual void compute(std::vector<POI2D>& poi_queue) = 0;

    oc_hip_engine* handle() { return engine_; }
    // images on the device (uploaded, or borrowed from an engine that holds the same pair); what prepare() / compute() do first
    void ensureImages() { uploadIfNeeded(); }

protected:
    oc_hip_engine* engine_ = nullptr;
    int device_ = hipdetail::default_device();
    std::atomic<bool> images_dirty_{false};   // (compute(POI*) may be entered by many threads at once: the upload happens once)
    std::mutex upload_mu_;
    unsigned long long images_seq_ = 0;

    void uploadIfNeeded() {
        if (!engine_) throw std::string("engine not created");
        if (!images_dirty_.load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> lock(upload_mu_);
        if (!images_dirty_.load(std::memory_order_relaxed)) return;
        if (!ref_img || !tar_img) throw std::string("setImages() has not been called");
        if (ref_img->height != tar_img->height || ref_img->width != tar_img->width)
            throw std::string("reference and target image sizes differ");
        hipdetail::SnapshotRegistry& reg = hipdetail::SnapshotRegistry::get();
        const void* r = ref_img->eg_mat.data();
        const void* t = tar_img->eg_mat.data();
        if (!reg.borrow(engine_, r, t, ref_img->height, ref_img->width, 1, device_, images_seq_)) {
            hipdetail::check(oc_hip_set_images2d(engine_, ref_img->eg_mat.data(), tar_img->eg_mat.data(), ref_img->height,
                                                 ref_img->width, OC_HIP_COL_MAJOR, OC_HIP_HOST));
            reg.publish(engine_, r, t, ref_img->height, ref_img->width, 1, device_);
        }
        images_dirty_.store(false, std::memory_order_release);
    }
    void computeBatch(POI2D* pois, size_t n) {
        uploadIfNeeded();
        hipdetail::check(oc_hip_compute(engine_, pois, n, sizeof(POI2D), OC_HIP_HOST));
    }
    // compute(POI2D*): the C-ABI's combining single-POI entry point (concurrent callers share a launch)
    void computeOne(POI2D* poi) {
        uploadIfNeeded();
        hipdetail::check(oc_hip_compute_one(engine_, poi));
    }
};

class DVC {
public:
    Image3D* ref_img = nullptr;
    Image3D* tar_img = nullptr;
    int subset_radius_x = 0, subset_radius_y = 0, subset_radius_z = 0;
    int thread_number = 1;

    DVC() {}
    virtual ~DVC() {
        if (engine_) {
            hipdetail::SnapshotRegistry::get().forget(engine_);
            oc_hip_destroy(engine_);
        }
    }
    DVC(const DVC&) = delete;
    DVC& operator=(const DVC&) = delete;

    void setImages(Image3D& ref, Image3D& tar) {
        ref_img = &ref;
        tar_img = &tar;
        images_dirty_ = true;
        images_seq_ = hipdetail::SnapshotRegistry::get().stamp();
        if (engine_) hipdetail::SnapshotRegistry::get().forget(engine_);
    }
    void setSubset(int radius_x, int radius_y, int radius_z) {
        subset_radius_x = radius_x;
        subset_radius_y = radius_y;
        subset_radius_z = radius_z;
        if (engine_) hipdetail::check(oc_hip_set_subset(engine_, radius_x, radius_y, radius_z));
    }
    void setDevice(int device) { setDevices(std::vector<int>(1, device)); }
    void setDevices(const std::vector<int>& devices) {
        if (devices.empty()) throw std::string("setDevices: empty device list");
        hipdetail::SnapshotRegistry::get().forget(engine_);
        hipdetail::check(oc_hip_set_devices(engine_, devices.data(), (int)devices.size()));
        device_ = devices[0];
        images_dirty_ = ref_img != nullptr;
    }

    virtual void prepare() = 0;
    virtual void compute(POI3D* poi) = 0;
    virtual void compute(std::vector<POI3D>& poi_queue) = 0;

    oc_hip_engine* handle() { return engine_; }
    void ensureImages() { uploadIfNeeded(); }

protected:
    oc_hip_engine* engine_ = nullptr;
    int device_ = hipdetail::default_device();
    std::atomic<bool> images_dirty_{false};   // (compute(POI*) may be entered by many threads at once: the upload happens once)
    std::mutex upload_mu_;
    unsigned long long images_seq_ = 0;

    void uploadIfNeeded() {
        if (!engine_) throw std::string("engine not created");
        if (!images_dirty_.load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> lock(upload_mu_);
        if (!images_dirty_.load(std::memory_order_relaxed)) return;
        if (!ref_img || !tar_img) throw std::string("setImages() has not been called");
        hipdetail::SnapshotRegistry& reg = hipdetail::SnapshotRegistry::get();
        const void* r = &ref_img->vol_mat[0][0][0];
        const void* t = &tar_img->vol_mat[0][0][0];
        if (!reg.borrow(engine_, r, t, ref_img->dim_x, ref_img->dim_y, ref_img->dim_z, device_, images_seq_)) {
            hipdetail::check(oc_hip_set_images3d(engine_, &ref_img->vol_mat[0][0][0], &tar_img->vol_mat[0][0][0],
                                                 ref_img->dim_x, ref_img->dim_y, ref_img->dim_z, OC_HIP_HOST));
            reg.publish(engine_, r, t, ref_img->dim_x, ref_img->dim_y, ref_img->dim_z, device_);
        }
        images_dirty_.store(false, std::memory_order_release);
    }
    void computeBatch(POI3D* pois, size_t n) {
        uploadIfNeeded();
        hipdetail::check(oc_hip_compute(engine_, pois, n, sizeof(POI3D), OC_HIP_HOST));
    }
    void computeOne(POI3D* poi) {
        uploadIfNeeded();
        hipdetail::check(oc_hip_compute_one(engine_, poi));
    }
};

// ---- FFT-accelerated cross correlation (integer-pixel initial guess) -----------------------------
class FFTCC2D : public DIC {
public:
    FFTCC2D(int subset_radius_x_, int subset_radius_y_, int thread_number_) {
        subset_radius_x = subset_radius_x_;
        subset_radius_y = subset_radius_y_;
        thread_number = thread_number_;
        hipdetail::check(oc_hip_fftcc2d_create(subset_radius_x_, subset_radius_y_, device_, &engine_));
        hipdetail::apply_default_devices(engine_);
    }
    void prepare() override {}  // empty in the reference too (src/oc_fftcc.cpp:175)
    void compute(POI2D* poi) override { computeOne(poi); }
    void compute(std::vector<POI2D>& poi_queue) override { computeBatch(poi_queue.data(), poi_queue.size()); }
};

class FFTCC3D : public DVC {
public:
    FFTCC3D(int rx, int ry, int rz, int thread_number_) {
        subset_radius_x = rx;
        subset_radius_y = ry;
        subset_radius_z = rz;
        thread_number = thread_number_;
        hipdetail::check(oc_hip_fftcc3d_create(rx, ry, rz, device_, &engine_));
        hipdetail::apply_default_devices(engine_);
    }
    void prepare() override {}
    void compute(POI3D* poi) override { computeOne(poi); }
    void compute(std::vector<POI3D>& poi_queue) override { computeBatch(poi_queue.data(), poi_queue.size()); }
};

// ---- inverse-compositional Gauss-Newton ---------------------------------------------------------------
template <class Base, class Poi>
class IcgnShim : public Base {
public:
    void setIteration(float conv_criterion_, float stop_condition_) {
        conv_criterion = conv_criterion_;
        stop_condition = stop_condition_;
        hipdetail::check(oc_hip_set_iteration(this->engine_, conv_criterion_, stop_condition_));
    }
    // setIteration(POI*): conv = poi->result.convergence, stop = (int)poi->result.iteration
    // (src/oc_icgn.cpp:109-113)
    void setIteration(Poi* poi) { setIteration(poi->result.convergence, (float)(int)poi->result.iteration); }

    void prepareRef() {
        this->uploadIfNeeded();
        hipdetail::check(oc_hip_prepare_ref(this->engine_));
    }
    void prepareTar() {
        this->uploadIfNeeded();
        hipdetail::check(oc_hip_prepare_tar(this->engine_));
    }
    void prepare() override {
        prepareRef();
        prepareTar();
    }
    // (callers inside their own OpenMP region -- src/oc_epipolar_search.cpp:184-188 -- are combined into one launch per batch)
    void compute(Poi* poi) override { this->computeOne(poi); }
    void compute(std::vector<Poi>& poi_queue) override { this->computeBatch(poi_queue.data(), poi_queue.size()); }
    // initial guess and refinement in one call and ONE round trip of the queue over PCIe: `first` (an FFTCC engine, say)
    // processes every POI, then this engine -- the same bits as first.compute(poi_queue); compute(poi_queue);
    void compute(std::vector<Poi>& poi_queue, Base& first) {
        this->ensureImages();
        first.ensureImages();
        if (!hipdetail::single_device(this->engine_) || !hipdetail::single_device(first.handle())) {
            first.compute(poi_queue);
            compute(poi_queue);
            return;
        }
        oc_hip_engine* chain[2] = {first.handle(), this->engine_};
        hipdetail::check(oc_hip_compute_chain(chain, 2, poi_queue.data(), poi_queue.size(), sizeof(Poi), OC_HIP_HOST));
    }

protected:
    float conv_criterion = 0.001f;
    float stop_condition = 10.f;
};

// the 2D extras of src/oc_icgn.h:75-76,130-131: centre-offset overloads and self-adaptive subsets
class Icgn2DShim : public IcgnShim<DIC, POI2D> {
public:
    using IcgnShim<DIC, POI2D>::compute;
    void setSelfAdaptive(bool is_self_adaptive) override {
        hipdetail::check(oc_hip_set_self_adaptive(engine_, is_self_adaptive ? 1 : 0));
        self_adaptive = is_self_adaptive;
    }
    void compute(POI2D* poi, Point2D& center_offset) {
        uploadIfNeeded();
        const float off[2] = {center_offset.x, center_offset.y};
        hipdetail::check(oc_hip_compute_one_with_offset(engine_, poi, off));
    }
    // Batched form of "refine several trial guesses per POI, keep the one with the highest ZNCC"
    // (EpipolarSearch::compute, src/oc_epipolar_search.cpp:150-190): `candidates` holds the trials of all POIs,
    // those of poi_queue[s] at [segment_starts[s], segment_starts[s+1]).  One ICGN launch, one selection kernel.
    void computeBestOf(std::vector<POI2D>& candidates, const std::vector<unsigned>& segment_starts, std::vector<POI2D>& poi_queue) {
        if (segment_starts.size() != poi_queue.size() + 1) throw std::string("computeBestOf: need one segment start per POI plus the end");
        compute(candidates);
        hipdetail::check(oc_hip_select_best(engine_, candidates.data(), candidates.size(), sizeof(POI2D), segment_starts.data(),
                                            poi_queue.size(), poi_queue.data(), sizeof(POI2D), OC_HIP_HOST));
    }
    void compute(std::vector<POI2D>& poi_queue, std::vector<Point2D>& center_offset_queue) {
        if (center_offset_queue.size() < poi_queue.size()) throw std::string("center_offset_queue is shorter than poi_queue");
        uploadIfNeeded();
        static_assert(sizeof(Point2D) == 2 * sizeof(float), "Point2D must be two packed floats");
        hipdetail::check(oc_hip_compute_with_offsets(engine_, poi_queue.data(),
                                                     reinterpret_cast<const float*>(center_offset_queue.data()),
                                                     poi_queue.size(), sizeof(POI2D), OC_HIP_HOST));
    }
};

class ICGN2D1 : public Icgn2DShim {
public:
    ICGN2D1(int rx, int ry, float conv_criterion_, float stop_condition_, int thread_number_) {
        subset_radius_x = rx;
        subset_radius_y = ry;
        conv_criterion = conv_criterion_;
        stop_condition = stop_condition_;
        thread_number = thread_number_;
        hipdetail::check(oc_hip_icgn2d1_create(rx, ry, conv_criterion_, stop_condition_, device_, &engine_));
        hipdetail::apply_default_devices(engine_);
    }
};

class ICGN2D2 : public Icgn2DShim {
public:
    ICGN2D2(int rx, int ry, float conv_criterion_, float stop_condition_, int thread_number_) {
        subset_radius_x = rx;
        subset_radius_y = ry;
        conv_criterion = conv_criterion_;
        stop_condition = stop_condition_;
        thread_number = thread_number_;
        hipdetail::check(oc_hip_icgn2d2_create(rx, ry, conv_criterion_, stop_condition_, device_, &engine_));
        hipdetail::apply_default_devices(engine_);
    }
};

// NR2D1(int rx, int ry, float conv_criterion, float stop_condition, int thread_number)  src/oc_nr.h / src/oc_nr.cpp:75-91.
// prepare() builds the target gradients and the three interpolation tables (src/oc_nr.cpp:119-158).
class NR2D1 : public IcgnShim<DIC, POI2D> {
public:
    NR2D1(int rx, int ry, float conv_criterion_, float stop_condition_, int thread_number_) {
        subset_radius_x = rx;
        subset_radius_y = ry;
        conv_criterion = conv_criterion_;
        stop_condition = stop_condition_;
        thread_number = thread_number_;
        hipdetail::check(oc_hip_nr2d1_create(rx, ry, conv_criterion_, stop_condition_, device_, &engine_));
        hipdetail::apply_default_devices(engine_);
    }
    void prepare() override {
        uploadIfNeeded();
        hipdetail::check(oc_hip_prepare(engine_));
    }
};

// ICLM2D1 / ICLM2D2(int rx, int ry, float conv_criterion, float stop_condition, int thread_number)
// src/oc_iclm.h:56-85, 104-133: prepare()/compute() as ICGN2D*, plus setDamping and self-adaptive subsets.
template <int KIND>
class Iclm2DShim : public IcgnShim<DIC, POI2D> {
public:
    Iclm2DShim(int rx, int ry, float conv_criterion_, float stop_condition_, int thread_number_) {
        subset_radius_x = rx;
        subset_radius_y = ry;
        conv_criterion = conv_criterion_;
        stop_condition = stop_condition_;
        thread_number = thread_number_;
        hipdetail::check(KIND == OC_HIP_ICLM2D1 ? oc_hip_iclm2d1_create(rx, ry, conv_criterion_, stop_condition_, device_, &engine_)
                                                : oc_hip_iclm2d2_create(rx, ry, conv_criterion_, stop_condition_, device_, &engine_));
        hipdetail::apply_default_devices(engine_);
    }
    void setDamping(float lambda, float alpha, float beta) { hipdetail::check(oc_hip_set_damping(engine_, lambda, alpha, beta)); }
    void setSelfAdaptive(bool is_self_adaptive) override {
        hipdetail::check(oc_hip_set_self_adaptive(engine_, is_self_adaptive ? 1 : 0));
        self_adaptive = is_self_adaptive;
    }
};
using ICLM2D1 = Iclm2DShim<OC_HIP_ICLM2D1>;
using ICLM2D2 = Iclm2DShim<OC_HIP_ICLM2D2>;

class ICGN3D1 : public IcgnShim<DVC, POI3D> {
public:
    ICGN3D1(int rx, int ry, int rz, float conv_criterion_, float stop_condition_, int thread_number_) {
        subset_radius_x = rx;
        subset_radius_y = ry;
        subset_radius_z = rz;
        conv_criterion = conv_criterion_;
        stop_condition = stop_condition_;
        thread_number = thread_number_;
        hipdetail::check(oc_hip_icgn3d1_create(rx, ry, rz, conv_criterion_, stop_condition_, device_, &engine_));
        hipdetail::apply_default_devices(engine_);
    }
};

// ---- the two selections of the RegionFit -> re-ICGN loop -----------------------------------------------------------
// examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:214-260 sorts a finished queue into reliable / unreliable vectors
// and, after every RegionFit + ICGN round, moves the POIs that now pass back.  splitReliable / mergeRecovered do the
// classification and the order-preserving compactions on the GPU (oc_hip_split_reliable / oc_hip_merge_recovered);
// `engine` is any engine of the matching dimension (it supplies the device and the stream).
template <class Engine, class Poi>
inline void splitReliable(Engine& engine, const std::vector<Poi>& poi_queue, float zncc_threshold_low, float zncc_threshold_high,
                          float conv_criterion, std::vector<Poi>& pois_reliable, std::vector<Poi>& pois_unreliable,
                          std::vector<unsigned>& pois_unreliable_idx) {
    const size_t n = poi_queue.size(), r0 = pois_reliable.size();
    pois_unreliable.clear();
    pois_unreliable_idx.clear();
    if (n == 0) return;
    pois_reliable.resize(r0 + n, poi_queue[0]);  // (POI2D / POI3D have no default constructor, like the reference's)
    pois_unreliable.assign(n, poi_queue[0]);
    pois_unreliable_idx.assign(n, 0u);
    size_t nr = 0, nu = 0;
    hipdetail::check(oc_hip_split_reliable(engine.handle(), poi_queue.data(), n, sizeof(Poi), sizeof(Poi) == OC_HIP_POI2D_BYTES ? 2 : 3,
                                           zncc_threshold_low, zncc_threshold_high, conv_criterion, pois_reliable.data(), r0,
                                           pois_unreliable.data(), pois_unreliable_idx.data(), &nr, &nu, OC_HIP_HOST));
    pois_reliable.resize(r0 + nr, poi_queue[0]);
    pois_unreliable.resize(nu, poi_queue[0]);
    pois_unreliable_idx.resize(nu);
}
// returns the number of POIs that became reliable in this round
template <class Engine, class Poi>
inline size_t mergeRecovered(Engine& engine, std::vector<Poi>& poi_queue, std::vector<Poi>& pois_unreliable,
                             std::vector<unsigned>& pois_unreliable_idx, float zncc_threshold_high, float conv_criterion,
                             std::vector<Poi>& pois_reliable) {
    const size_t n = pois_unreliable.size(), r0 = pois_reliable.size();
    if (n == 0) return 0;
    pois_reliable.resize(r0 + n, pois_unreliable[0]);
    size_t nrec = 0, nrem = 0;
    hipdetail::check(oc_hip_merge_recovered(engine.handle(), poi_queue.data(), poi_queue.size(), sizeof(Poi), sizeof(Poi) == OC_HIP_POI2D_BYTES ? 2 : 3,
                                            pois_unreliable.data(), pois_unreliable_idx.data(), n, zncc_threshold_high, conv_criterion,
                                            pois_reliable.data(), r0, &nrec, &nrem, OC_HIP_HOST));
    pois_reliable.resize(r0 + nrec, pois_unreliable[0]);
    pois_unreliable.resize(nrem, pois_unreliable[0]);
    pois_unreliable_idx.resize(nrem);
    return nrec;
}

// ---- several engines, one queue, ONE round trip over PCIe --------------------------------------------------------
// The reference's mains call fftcc->compute(poi_queue); icgn->compute(poi_queue); back to back
// (examples/test_2d_dic_fftcc_icgn1.cpp:80-99, examples/test_dvc_fftcc_icgn1.cpp:87-106).  Through these shims each call
// moves the whole POI vector to the GPU and back.  computeChain({fftcc, icgn}, poi_queue) is the same computation -- bit
// for bit -- with one copy in and one copy out (oc_hip_compute_chain; 250 000 POIs: two crossings of 25 MB instead of
// four).  Engines spread over several GPUs (setDevices / OC_HIP_DEVICES) run one after the other as before.
namespace hipdetail {
template <class Engine, class Poi>
inline void compute_chain(const std::vector<Engine*>& engines, std::vector<Poi>& poi_queue) {
    if (engines.empty()) throw std::string("computeChain: no engine given");
    std::vector<oc_hip_engine*> handles;
    bool plain = true;
    for (Engine* e : engines) {
        if (!e) throw std::string("computeChain: null engine");
        e->ensureImages();
        handles.push_back(e->handle());
        plain = plain && single_device(e->handle());
    }
    if (!plain) {
        for (Engine* e : engines) e->compute(poi_queue);
        return;
    }
    check(oc_hip_compute_chain(handles.data(), (int)handles.size(), poi_queue.data(), poi_queue.size(), sizeof(Poi), OC_HIP_HOST));
}
}  // namespace hipdetail
inline void computeChain(const std::vector<DIC*>& engines, std::vector<POI2D>& poi_queue) { hipdetail::compute_chain(engines, poi_queue); }
inline void computeChain(const std::vector<DVC*>& engines, std::vector<POI3D>& poi_queue) { hipdetail::compute_chain(engines, poi_queue); }


// Strain(float subregion_radius, int neighbor_number_min, int thread_number)  src/oc_strain.h:34-73.
// prepare(poi_queue) builds the neighbour search over the queue's coordinates, compute(poi_queue) writes
// poi.strain of every POI that can be fitted (src/oc_strain.cpp:96-147, 236-247, 476-488).  The POI2DS (stereo)
// overloads are not offered.
class Strain {
public:
    Strain(float subregion_radius_, int neighbor_number_min_, int thread_number_)
        : subregion_radius(subregion_radius_), neighbor_number_min(neighbor_number_min_), thread_number(thread_number_) {
        hipdetail::check(oc_hip_strain_create(subregion_radius_, neighbor_number_min_, hipdetail::default_device(), &engine_));
    }
    ~Strain() { if (engine_) oc_hip_destroy(engine_); }
    Strain(const Strain&) = delete;
    Strain& operator=(const Strain&) = delete;

    float getSubregionRadius() const { return subregion_radius; }
    int getNeighborMin() const { return neighbor_number_min; }
    float getZnccThreshold() const { return zncc_threshold; }
    void setSubregionRadius(float v) { push(v, neighbor_number_min, zncc_threshold, approximation); }
    void setNeighborMin(int v) { push(subregion_radius, v, zncc_threshold, approximation); }
    void setZnccThreshold(float v) { push(subregion_radius, neighbor_number_min, v, approximation); }
    void setDescription(int v) { description = v; }  // stored; the reference's compute() does not read it either
    void setApproximation(int v) { push(subregion_radius, neighbor_number_min, zncc_threshold, v); }  // 1 Cauchy, 2 Green

    void prepare(std::vector<POI2D>& q) { hipdetail::check(oc_hip_strain_prepare(engine_, q.data(), q.size(), sizeof(POI2D), 2, OC_HIP_HOST)); }
    void prepare(std::vector<POI3D>& q) { hipdetail::check(oc_hip_strain_prepare(engine_, q.data(), q.size(), sizeof(POI3D), 3, OC_HIP_HOST)); }
    void compute(std::vector<POI2D>& q) { hipdetail::check(oc_hip_strain_compute(engine_, q.data(), q.size(), sizeof(POI2D), 2, OC_HIP_HOST)); }
    void compute(std::vector<POI3D>& q) { hipdetail::check(oc_hip_strain_compute(engine_, q.data(), q.size(), sizeof(POI3D), 3, OC_HIP_HOST)); }
    // compute(POI*, poi_queue) (src/oc_strain.cpp:149, :372): `poi` must be an element of poi_queue; the whole queue is
    // evaluated on a copy and only that POI's strain is taken over (use the queue overload for more than a few POIs)
    void compute(POI2D* poi, std::vector<POI2D>& q) { one(poi, q, 2); }
    void compute(POI3D* poi, std::vector<POI3D>& q) { one(poi, q, 3); }

protected:
    float subregion_radius;
    int neighbor_number_min;
    float zncc_threshold = 0.9f;
    int description = 1;
    int approximation = 1;
    int thread_number;

private:
    oc_hip_engine* engine_ = nullptr;
    void push(float r, int n, float z, int a) {
        hipdetail::check(oc_hip_strain_set(engine_, r, n, z, a));
        subregion_radius = r;
        neighbor_number_min = n;
        zncc_threshold = z;
        approximation = a;
    }
    template <class Poi>
    void one(Poi* poi, std::vector<Poi>& q, int ndim) {
        if (q.empty() || poi < q.data() || poi >= q.data() + q.size()) throw std::string("Strain::compute(POI*, queue): the POI must be an element of the queue");
        std::vector<Poi> copy(q);
        hipdetail::check(oc_hip_strain_compute(engine_, copy.data(), copy.size(), sizeof(Poi), ndim, OC_HIP_HOST));
        poi->strain = copy[poi - q.data()].strain;
    }
};

// RegionFit2D / RegionFit3D(float neighbor_search_radius, int neighbor_number_min, int thread_number)
// src/oc_region_fit.h:28-80: setNeighbor(reliable) ; prepare() ; compute(poi_queue).  The reference derives them from
// DIC / DVC without using the images; here they are plain classes.
template <class Poi, int NDIM>
class RegionFitShim {
public:
    RegionFitShim(float neighbor_search_radius_, int neighbor_number_min_, int thread_number_)
        : neighbor_search_radius(neighbor_search_radius_), neighbor_number_min(neighbor_number_min_), thread_number(thread_number_) {
        hipdetail::check(oc_hip_region_fit_create(neighbor_search_radius_, neighbor_number_min_, hipdetail::default_device(), &engine_));
    }
    ~RegionFitShim() { if (engine_) oc_hip_destroy(engine_); }
    RegionFitShim(const RegionFitShim&) = delete;
    RegionFitShim& operator=(const RegionFitShim&) = delete;

    float getSearchRadius() const { return neighbor_search_radius; }
    int getNeighborMin() const { return neighbor_number_min; }
    void setSearchRadius(float v) {
        hipdetail::check(oc_hip_region_fit_set(engine_, v, neighbor_number_min));
        neighbor_search_radius = v;
    }
    void setNeighborMin(int v) {
        hipdetail::check(oc_hip_region_fit_set(engine_, neighbor_search_radius, v));
        neighbor_number_min = v;
    }
    void setNeighbor(std::vector<Poi>& reliable_pois) { neighbor_reliable = &reliable_pois; }
    void prepare() {
        if (!neighbor_reliable) throw std::string("RegionFit::prepare: setNeighbor has not been called");
        hipdetail::check(oc_hip_region_fit_prepare(engine_, neighbor_reliable->data(), neighbor_reliable->size(), sizeof(Poi), NDIM, OC_HIP_HOST));
    }
    void compute(Poi* poi) { hipdetail::check(oc_hip_region_fit_compute(engine_, poi, 1, sizeof(Poi), NDIM, OC_HIP_HOST)); }
    void compute(std::vector<Poi>& q) { hipdetail::check(oc_hip_region_fit_compute(engine_, q.data(), q.size(), sizeof(Poi), NDIM, OC_HIP_HOST)); }

protected:
    std::vector<Poi>* neighbor_reliable = nullptr;
    float neighbor_search_radius;
    int neighbor_number_min;
    int thread_number;

private:
    oc_hip_engine* engine_ = nullptr;
};
using RegionFit2D = RegionFitShim<POI2D, 2>;
using RegionFit3D = RegionFitShim<POI3D, 3>;

// ---- the reference's CUDA-module shapes (gpu_lib/opencorr_gpu.h:31-101), so that
// examples/test_2d_dic_gpu_icgn.cpp / test_dvc_gpu_icgn.cpp compile against this header --------------
struct Img2D { int width, height; float* data; };           // row-major
struct Img3D { int dim_x, dim_y, dim_z; float* data; };     // z, y, x contiguous

template <int KIND>
class IcgnGpu2D {
    oc_hip_engine* e_ = nullptr;

public:
    IcgnGpu2D(int rx, int ry, float conv, int stop) {
        hipdetail::check(KIND == OC_HIP_ICGN2D1 ? oc_hip_icgn2d1_create(rx, ry, conv, (float)stop, hipdetail::default_device(), &e_)
                                                : oc_hip_icgn2d2_create(rx, ry, conv, (float)stop, hipdetail::default_device(), &e_));
        hipdetail::apply_default_devices(e_);
    }
    ~IcgnGpu2D() { if (e_) oc_hip_destroy(e_); }
    IcgnGpu2D(const IcgnGpu2D&) = delete;
    IcgnGpu2D& operator=(const IcgnGpu2D&) = delete;
    void setImages(Img2D ref, Img2D tar) {
        hipdetail::check(oc_hip_set_images2d(e_, ref.data, tar.data, ref.height, ref.width, OC_HIP_ROW_MAJOR, OC_HIP_HOST));
    }
    void setSubset(int rx, int ry) { hipdetail::check(oc_hip_set_subset(e_, rx, ry, 0)); }
    void setIteration(float conv, int stop) { hipdetail::check(oc_hip_set_iteration(e_, conv, (float)stop)); }
    void prepare() { hipdetail::check(oc_hip_prepare(e_)); }
    void compute(std::vector<POI2D>& q) { hipdetail::check(oc_hip_compute(e_, q.data(), q.size(), sizeof(POI2D), OC_HIP_HOST)); }
};
typedef IcgnGpu2D<OC_HIP_ICGN2D1> ICGN2D1GPU;
typedef IcgnGpu2D<OC_HIP_ICGN2D2> ICGN2D2GPU;

class ICGN3D1GPU {
    oc_hip_engine* e_ = nullptr;

public:
    ICGN3D1GPU(int rx, int ry, int rz, float conv, int stop) {
        hipdetail::check(oc_hip_icgn3d1_create(rx, ry, rz, conv, (float)stop, hipdetail::default_device(), &e_));
        hipdetail::apply_default_devices(e_);
    }
    ~ICGN3D1GPU() { if (e_) oc_hip_destroy(e_); }
    ICGN3D1GPU(const ICGN3D1GPU&) = delete;
    ICGN3D1GPU& operator=(const ICGN3D1GPU&) = delete;
    void setImages(Img3D ref, Img3D tar) {
        hipdetail::check(oc_hip_set_images3d(e_, ref.data, tar.data, ref.dim_x, ref.dim_y, ref.dim_z, OC_HIP_HOST));
    }
    void setSubset(int rx, int ry, int rz) { hipdetail::check(oc_hip_set_subset(e_, rx, ry, rz)); }
    void setIteration(float conv, int stop) { hipdetail::check(oc_hip_set_iteration(e_, conv, (float)stop)); }
    void prepare() { hipdetail::check(oc_hip_prepare(e_)); }
    void compute(std::vector<POI3D>& q) { hipdetail::check(oc_hip_compute(e_, q.data(), q.size(), sizeof(POI3D), OC_HIP_HOST)); }
};

}  // namespace opencorr
