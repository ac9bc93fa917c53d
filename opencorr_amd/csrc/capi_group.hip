// capi_group.hip -- device groups: block cuts, the RCCL binding, group compute paths (part of the C-ABI of include/opencorr_hip.h; split from capi.hip in round 6, same exported symbols)
#include "capi_internal.h"

namespace ochip_capi {

// ---------------------------------------------------------------------------
// device groups (oc_hip_set_devices): contiguous blocks of the queue, one per member (the loop that is
// being replaced is src/oc_icgn.cpp:343-351).  Member g takes POIs [g * ceil(n / G), (g + 1) * ceil(n / G)).
// ---------------------------------------------------------------------------
struct GroupBlock {
    oc_hip_engine* e;
    size_t first, n;
};

std::vector<GroupBlock> group_blocks(oc_hip_engine* e, size_t count) {
    const size_t G = e->replicas.size() + 1, per = (count + G - 1) / G;
    std::vector<GroupBlock> b;
    for (size_t g = 0; g < G; g++) {
        const size_t first = std::min(g * per, count), n = std::min(per, count - first);
        b.push_back({g == 0 ? e : e->replicas[g - 1], first, n});
    }
    return b;
}

// RCCL, loaded on first use: the single-GPU path never needs it and should not pay for loading it (nor fail to start
// on a machine without it).  Types and enumerators come from <rccl/rccl.h> at compile time -- the datatype passed to
// ncclAllGather is the header's ncclUint8, not a number typed in here -- and the loaded library must report the same
// major version as that header.
struct Rccl {
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) err = nullptr;
    decltype(&ncclGetVersion) get_version = nullptr;
    int version = 0;
    std::string why;  // why the library is unusable
    bool ok = false;
    const char* text(ncclResult_t rc) const { return err ? err(rc) : "?"; }
    static Rccl& get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, [] {
            // OC_HIP_RCCL_LIB names a specific build; otherwise the ROCm soname, then the unversioned name
            const char* names[3] = {getenv("OC_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so"};
            void* h = nullptr;
            for (const char* name : names)
                if (!h && name && *name) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                const char* d = dlerror();
                r.why = std::string("librccl.so.1 could not be loaded: ") + (d ? d : "?");
                return;
            }
            r.comm_init_all = (decltype(r.comm_init_all))dlsym(h, "ncclCommInitAll");
            r.all_gather = (decltype(r.all_gather))dlsym(h, "ncclAllGather");
            r.group_start = (decltype(r.group_start))dlsym(h, "ncclGroupStart");
            r.group_end = (decltype(r.group_end))dlsym(h, "ncclGroupEnd");
            r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
            r.err = (decltype(r.err))dlsym(h, "ncclGetErrorString");
            r.get_version = (decltype(r.get_version))dlsym(h, "ncclGetVersion");
            if (!(r.comm_init_all && r.all_gather && r.group_start && r.group_end && r.comm_destroy && r.get_version)) {
                r.why = "librccl lacks one of ncclCommInitAll / ncclAllGather / ncclGroupStart / ncclGroupEnd / ncclCommDestroy / ncclGetVersion";
                return;
            }
            if (r.get_version(&r.version) != ncclSuccess || r.version / 10000 != NCCL_MAJOR) {
                r.why = "librccl reports version " + std::to_string(r.version) + ", built against major " + std::to_string(NCCL_MAJOR);
                return;
            }
            r.ok = true;
        });
        return r;
    }
};

void group_drop_comms(oc_hip_engine* e) {
    bool any = e->rccl_comm != nullptr;
    for (oc_hip_engine* r : e->replicas) any = any || r->rccl_comm != nullptr;
    if (!any) return;  // never load RCCL just to find out there is nothing to drop
    Rccl& R = Rccl::get();
    auto drop = [&](oc_hip_engine* m) {
        if (m->rccl_comm && R.ok) {
            (void)hipSetDevice(m->device);
            (void)R.comm_destroy((ncclComm_t)m->rccl_comm);
        }
        m->rccl_comm = nullptr;
    };
    drop(e);
    for (oc_hip_engine* r : e->replicas) drop(r);
}

// Can this group's all-gather be ONE ncclAllGather?  Yes when its members sit on distinct devices (a communicator
// cannot hold a device twice) and there is more than one of them -- or exactly one and "group_force_rccl" is set, which
// runs the identical code (ncclCommInitAll over one device, ncclAllGather on a one-rank communicator) so that the RCCL
// binding executes on a one-GPU machine.
bool group_uses_rccl(oc_hip_engine* e, const std::vector<GroupBlock>& blocks) {
    const int G = (int)blocks.size();
    for (int a = 0; a < G; a++)
        for (int b = a + 1; b < G; b++)
            if (blocks[a].e->device == blocks[b].e->device) return false;
    if (G == 1 && !e->group_force_rccl) return false;
    return true;
}

// Every member's mirror ends up holding the whole queue (blocks of `block` bytes, the last one padded): ONE
// ncclAllGather over xGMI when the members sit on distinct devices, peer copies otherwise (a group may name a device
// twice -- that is how the sharding logic is exercised on a one-GPU box).  RCCL: every member sends its own block from
// where it lies -- the leader straight from the caller's queue (`leader_block`), the others from their slot of their
// mirror (an in-place all-gather for them).
int group_allgather(oc_hip_engine* e, const std::vector<GroupBlock>& blocks, size_t block, const char* leader_block) {
    const int G = (int)blocks.size();
    if (group_uses_rccl(e, blocks)) {
        Rccl& R = Rccl::get();
        if (!R.ok) {
            if (e->group_force_rccl) return fail(OC_HIP_ERR_HIP, "group_force_rccl: %s", R.why.c_str());
        } else {
            if (!e->rccl_comm) {
                std::vector<ncclComm_t> comms(G, nullptr);
                std::vector<int> devs;
                for (const GroupBlock& b : blocks) devs.push_back(b.e->device);
                const ncclResult_t rc = R.comm_init_all(comms.data(), G, devs.data());
                (void)hipSetDevice(e->device);
                if (rc != ncclSuccess) return fail(OC_HIP_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", G, R.text(rc));
                for (int g = 0; g < G; g++) blocks[g].e->rccl_comm = comms[g];
            }
            // whatever happens between ncclGroupStart and ncclGroupEnd, ncclGroupEnd is reached: an open group would
            // poison this thread's later RCCL calls and the cached communicators
            ncclResult_t rc = R.group_start();
            hipError_t herr = hipSuccess;
            if (rc == ncclSuccess) {
                for (int g = 0; g < G && rc == ncclSuccess && herr == hipSuccess; g++) {
                    oc_hip_engine* m = blocks[g].e;
                    herr = hipSetDevice(m->device);
                    if (herr != hipSuccess) break;
                    char* mirror = m->group_mirror.as<char>();
                    const char* mine = g == 0 ? leader_block : mirror + (size_t)g * block;
                    rc = R.all_gather(mine, mirror, block, ncclUint8, (ncclComm_t)m->rccl_comm, m->stream);
                }
                const ncclResult_t rc2 = R.group_end();
                if (rc == ncclSuccess) rc = rc2;
            }
            (void)hipSetDevice(e->device);
            if (herr != hipSuccess) return fail(OC_HIP_ERR_HIP, "hipSetDevice inside the all-gather failed: %s", hipGetErrorString(herr));
            if (rc != ncclSuccess) return fail(OC_HIP_ERR_HIP, "ncclAllGather failed: %s", R.text(rc));
            return OC_HIP_OK;
        }
    }
    // peer copies: the leader's block joins its mirror, then member g fetches every other member's block once that
    // member is done (its event)
    OC_HIP_TRY(hipMemcpyAsync(e->group_mirror.p, leader_block, block, hipMemcpyDeviceToDevice, e->stream));  // blocks[0] is a full block
    OC_HIP_TRY(hipEventRecord(e->group_ev, e->stream));
    for (int g = 0; g < G; g++) {
        oc_hip_engine* m = blocks[g].e;
        OC_HIP_TRY(hipSetDevice(m->device));
        for (int h = 0; h < G; h++) {
            if (h == g) continue;
            oc_hip_engine* src = blocks[h].e;
            OC_HIP_TRY(hipStreamWaitEvent(m->stream, src->group_ev, 0));
            OC_HIP_TRY(hipMemcpyPeerAsync(m->group_mirror.as<char>() + (size_t)h * block, m->device,
                                          src->group_mirror.as<char>() + (size_t)h * block, src->device, block, m->stream));
        }
    }
    OC_HIP_TRY(hipSetDevice(e->device));
    return OC_HIP_OK;
}

// DEVICE queue of a group: the queue lives on the leader's device.  Every other member pulls its block into its own
// mirror (peer copy over xGMI), solves it there on its own stream and pushes the records back; the leader solves
// block 0 in place.  All of it is stream-ordered: members wait for the leader's stream to reach this call, the
// leader's stream waits for the members' completion events.
int compute_group_device(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes) {
    const std::vector<GroupBlock> blocks = group_blocks(e, count);
    const int stride_f = (int)(stride_bytes / 4);
    const size_t per = blocks[0].n, block = per * stride_bytes;  // blocks[0] is the largest
    auto event_of = [](oc_hip_engine* m) -> int {
        if (!m->group_ev) OC_HIP_TRY(hipEventCreateWithFlags(&m->group_ev, hipEventDisableTiming));
        return OC_HIP_OK;
    };
    OC_TRY(event_of(e));
    OC_HIP_TRY(hipEventRecord(e->group_ev, e->stream));  // the queue is ready when the leader's stream gets here
    hipEvent_t ready = e->group_ev;
    for (size_t g = 1; g < blocks.size(); g++) {
        oc_hip_engine* m = blocks[g].e;
        const size_t first = blocks[g].first, n = blocks[g].n;
        std::lock_guard<std::mutex> lock(m->mu);
        OC_HIP_TRY(hipSetDevice(m->device));
        OC_TRY(event_of(m));
        OC_TRY(m->group_mirror.reserve(blocks.size() * block));
        OC_HIP_TRY(hipStreamWaitEvent(m->stream, ready, 0));
        if (n) {
            char* mine = m->group_mirror.as<char>() + g * block;
            OC_HIP_TRY(hipMemcpyPeerAsync(mine, m->device, pois + first * stride_bytes, e->device, n * stride_bytes, m->stream));
            const float* d_off = nullptr;
            if (offsets) {
                OC_TRY(m->group_off_mirror.reserve(per * 2 * sizeof(float)));
                OC_HIP_TRY(hipMemcpyPeerAsync(m->group_off_mirror.p, m->device, offsets + 2 * first, e->device, n * 2 * sizeof(float), m->stream));
                d_off = m->group_off_mirror.as<float>();
            }
            OC_TRY(run_compute_device(m, reinterpret_cast<float*>(mine), stride_f, n, d_off));
            OC_HIP_TRY(hipMemcpyPeerAsync(pois + first * stride_bytes, e->device, mine, m->device, n * stride_bytes, m->stream));
        }
        OC_HIP_TRY(hipEventRecord(m->group_ev, m->stream));
    }
    OC_HIP_TRY(hipSetDevice(e->device));
    // the members are busy; now the leader's own block, in place
    if (blocks[0].n) OC_TRY(run_compute_device(e, reinterpret_cast<float*>(pois), stride_f, blocks[0].n, offsets));
    if (e->group_allgather) {
        // one all-gather: every member ends up with every block
        OC_TRY(e->group_mirror.reserve(blocks.size() * block));
        OC_TRY(group_allgather(e, blocks, block, pois));
        for (const GroupBlock& b : blocks) {
            b.e->group_mirror_block = block;
            if (b.e != e) {
                OC_HIP_TRY(hipSetDevice(b.e->device));
                OC_HIP_TRY(hipEventRecord(b.e->group_ev, b.e->stream));
            }
        }
        OC_HIP_TRY(hipSetDevice(e->device));
    }
    for (size_t g = 1; g < blocks.size(); g++) OC_HIP_TRY(hipStreamWaitEvent(e->stream, blocks[g].e->group_ev, 0));
    return OC_HIP_OK;
}

// HOST queue of a group: every member moves and solves its own block (its own host thread, its own PCIe link), the
// results land directly in the caller's vector -- no exchange step is needed for a host-resident queue.
int compute_group_host(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes) {
    const std::vector<GroupBlock> blocks = group_blocks(e, count);
    std::vector<int> rc(blocks.size(), OC_HIP_OK);
    std::vector<std::string> msg(blocks.size());
    auto work = [&](size_t g) {
        oc_hip_engine* m = blocks[g].e;
        if (blocks[g].n == 0) return;
        if (hipSetDevice(m->device) != hipSuccess) {
            rc[g] = OC_HIP_ERR_HIP;
            msg[g] = "hipSetDevice failed";
            return;
        }
        rc[g] = compute_host(m, pois + blocks[g].first * stride_bytes, offsets ? offsets + 2 * blocks[g].first : nullptr, blocks[g].n,
                             stride_bytes);
        if (rc[g] != OC_HIP_OK) msg[g] = g_last_error;  // thread-local in the worker
    };
    std::vector<std::thread> threads;
    for (size_t g = 1; g < blocks.size(); g++) threads.emplace_back([&, g] {
        std::lock_guard<std::mutex> lock(blocks[g].e->mu);
        work(g);
    });
    work(0);
    for (std::thread& t : threads) t.join();
    OC_HIP_TRY(hipSetDevice(e->device));
    for (size_t g = 0; g < blocks.size(); g++)
        if (rc[g] != OC_HIP_OK) return fail(rc[g], "group member %zu (device %d): %s", g, blocks[g].e->device, msg[g].c_str());
    return OC_HIP_OK;
}


}  // namespace ochip_capi
