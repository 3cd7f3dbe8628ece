// CloudSliceProcessor.h -- the backend thread right behind the tracker (backend/CloudSliceProcessor.h:34-51, .cpp:38-231) against this
// shell: it takes every CloudSlice the tracker hands over (cloudMutex / cloudSignal / cycledMutex protocol, the FIRST pseudo-slice,
// latestPoseId), fills CloudSlice::processedCloud -- weight cull, pcl::VoxelGrid at the voxel size, 20-NN normals: ONE call of
// kt_slice_process on the GPU instead of three PCL passes on the CPU -- and, at the end of a run, writes what the reference's save()
// writes: the concatenated processed clouds, voxel-gridded once more unless overlaps are kept, as a binary PCD of
// pcl::PointXYZRGBNormal (kt_host_voxel_grid_normal / kt_host_save_pcd).
// A ThreadObject like the reference's: a controller starts it with std::thread(&ThreadObject::start, processor); it ends itself once it
// has taken over the FINAL slice (the tracker thread keeps waking it until then, TrackerInterface.cpp:66-69).
#pragma once

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "ConfigArgs.h"
#include "KintinuousTracker.h"
#include "ThreadObject.h"
#include "Volume.h"

class CloudSliceProcessor : public ThreadObject {
  public:
    CloudSliceProcessor() : ThreadObject("CloudSliceProcessorThread"), ctx(0) { reset(); }
    virtual ~CloudSliceProcessor()
    {
        if (ctx) kt_ctx_destroy(ctx);
    }

    void reset()
    {
        latestPushedCloud = 0;
        cycledMutex = false;
    }

    // :180-231.  Returns the number of points written (the reference prints it), or -1 when the file cannot be written.
    long long save()
    {
        if (!(threadPack.finalised.getValue() && threadPack.cloudSlices.size() > 1)) return -1;   // assert(finalised && cloudSlices.size() > 1), :182
        CloudSlice::PointCloudNormal fullCloud;
        const int latestPoseIdCopy = threadPack.latestPoseId.getValue();
        for (int i = 1; i < latestPoseIdCopy; i++)
            fullCloud.insert(fullCloud.end(), threadPack.cloudSlices.at(i)->processedCloud->begin(), threadPack.cloudSlices.at(i)->processedCloud->end());
        if (ConfigArgs::get().extractOverlap && !ConfigArgs::get().saveOverlap && fullCloud.size()) {
            CloudSlice::PointCloudNormal tempCloud(fullCloud.size());
            size_t kept = 0;
            ktSafeCall(kt_host_voxel_grid_normal(reinterpret_cast<const kt_point_xyzrgbnormal*>(fullCloud.data()), fullCloud.size(), leafSize(),
                                                 reinterpret_cast<kt_point_xyzrgbnormal*>(tempCloud.data()), &kept));
            tempCloud.resize(kept);
            fullCloud.swap(tempCloud);
        }
        std::printf("Saving %zu points... ", fullCloud.size());
        std::fflush(stdout);
        const std::string filePCD = ConfigArgs::get().saveFile + ".pcd";
        if (kt_host_save_pcd(filePCD.c_str(), reinterpret_cast<const kt_point_xyzrgbnormal*>(fullCloud.data()), fullCloud.size()) != KT_OK) {
            std::printf("failed: %s\n", kt_last_error());
            return -1;
        }
        std::printf("PCD saved\n");
        return (long long)fullCloud.size();
    }

  private:
    // one turn of ThreadObject::run()'s loop (CloudSliceProcessor.cpp:38-178); false once the FINAL slice has been taken over.
    // The wait on cloudSignal is bounded at 50 ms (the reference waits without a bound, :42, and relies on the tracker's per-frame
    // notify and on the end-of-run wake-ups of TrackerInterface.cpp:66-69; so does this, the bound only covers a lost wake-up).
    bool inline process()
    {
        const int wait_ms = 50;
        std::unique_lock<std::mutex> lock(threadPack.tracker->cloudMutex);
        threadPack.tracker->cloudSignal.wait_for(lock, std::chrono::milliseconds(wait_ms));
        std::vector<CloudSlice*>* trackerSlices = &threadPack.tracker->getCloudSlices();
        numClouds = (int)trackerSlices->size();
        cycledMutex = threadPack.tracker->cycledMutex;
        if (cycledMutex) threadPack.tracker->cycledMutex = false;

        if (threadPack.cloudSlices.size() == 0) {
            const uint64_t initTime = threadPack.tracker->init_utime.getValue();
            if (initTime == std::numeric_limits<unsigned long long>::max()) return true;   // no frame tracked yet
            kt::Matrix3f lastRotation = threadPack.tracker->getLastRotation();
            kt::Vector3f lastTranslation = threadPack.tracker->getLastTranslation();
            threadPack.cloudSlices.push_back(new CloudSlice(new CloudSlice::PointCloud(), CloudSlice::FIRST, CloudSlice::FAIL, lastTranslation,
                                                            lastRotation, initTime, Stopwatch::getCurrentSystemTime(), 0, 0, 0, 0, &threadPack.tracker->placeRecognitionBuffer[0]));
            threadPack.cloudSlices.back()->processedCloud = new CloudSlice::PointCloudNormal();
            threadPack.latestPoseId.assignAndNotifyAll((int)threadPack.cloudSlices.size());
        }
        lock.unlock();

        // (the reference enters this loop on cycledMutex only, :85; a slice that arrived while the previous turn was still working is
        // taken as well here -- with a bounded wait above nothing would wake the thread for it otherwise)
        while (latestPushedCloud < numClouds) {
            CloudSlice* s = trackerSlices->at(latestPushedCloud);
            // :89-160: weight cull (alpha >= -cw), VoxelGrid with leaf = the largest voxel edge, NormalEstimation k = 20,
            // concatenateFields -> processedCloud
            size_t np = 0;
            if (s->processedCloud) {
                // the tracker ran the stage on the device right behind the extraction kernel (KintinuousTracker::enableSliceStage):
                // the slab never left the GPU in between, and nothing is left to do here but the hand-over
                np = s->processedCloud->size();
            } else {
                s->processedCloud = new CloudSlice::PointCloudNormal(s->cloud->size());
                if (s->cloud->size()) {
                    static_assert(sizeof(PointXYZRGBNormal) == sizeof(kt_point_xyzrgbnormal), "processedCloud layout");
                    ktSafeCall(kt_slice_process(context(), s->cloud->data(), s->cloud->size(), ConfigArgs::get().weightCull, leafSize(), 20,
                                                reinterpret_cast<kt_point_xyzrgbnormal*>(s->processedCloud->data()), &np));
                }
                s->processedCloud->resize(np);
            }
            // the reference culls and down-samples slice->cloud IN PLACE (:112-114, :138-140): later backend threads see the voxel-gridded
            // points there too.  They are the processed cloud without its normals.
            s->cloud->resize(np);
            for (size_t i = 0; i < np; ++i) {
                const PointXYZRGBNormal& q = (*s->processedCloud)[i];
                PointXYZRGB& p = (*s->cloud)[i];
                std::memset(&p, 0, sizeof(p));
                p.x = q.x; p.y = q.y; p.z = q.z; p.pad0 = 1.0f;
                p.b = q.b; p.g = q.g; p.r = q.r; p.a = q.a;
            }
            threadPack.cloudSlices.push_back(s);
            threadPack.latestPoseId.assignAndNotifyAll((int)threadPack.cloudSlices.size());
            latestPushedCloud++;
        }
        if (latestPushedCloud) lagTime.assignValue(Stopwatch::getCurrentSystemTime() - trackerSlices->at(latestPushedCloud - 1)->lagTime);   // :165-168
        if (threadPack.cloudSlices.size() && threadPack.cloudSlices.back()->dimension == CloudSlice::FINAL) {
            threadPack.cloudSliceProcessorFinished.assignAndNotifyAll(true);
            lagTime.assignValue(0);
            return false;
        }
        return true;
    }

    static float leafSize()
    {
        const float3& v = Volume::get().getVoxelSizeMeters();
        return std::max(v.x, std::max(v.y, v.z));
    }
    // this thread's own context (its own stream): a kt_ctx is not shared between threads
    kt_ctx* context()
    {
        if (!ctx) ktSafeCall(kt_ctx_create(ConfigArgs::get().gpu, &ctx));
        return ctx;
    }

    int latestPushedCloud;
    int numClouds;
    bool cycledMutex;
    kt_ctx* ctx;
};
