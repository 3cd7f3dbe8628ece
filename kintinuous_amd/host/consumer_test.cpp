// consumer_test -- a stand-in for the reference's backend consumer thread: the tracker-facing half of CloudSliceProcessor::process
// (backend/CloudSliceProcessor.cpp:38-83, 163-181) run against this shell while the tracker thread plays a log.  It touches exactly
// what the reference's consumer touches -- tracker->cloudMutex, cloudSignal.wait(cloudMutex), getCloudSlices(), cycledMutex,
// init_utime, getLastRotation / getLastTranslation, placeRecognitionBuffer[0], CloudSlice's 12-argument constructor, processedCloud,
// lagTime, the FINAL dimension -- so compiling and finishing is the test.  Usage: consumer_test -l <log.klg> [tracker options] [-v x]
// prints: consumed <slices> first_utime <t> pr <samples> loops <n> poses <n> live <tsdf points> <image 0|1>
#include <atomic>
#include <cstdio>
#include <limits>
#include <thread>

#include "TrackerInterface.h"

struct ThreadDataPack {   // backend/ThreadDataPack.h: only what the cloud slice processor uses
    KintinuousTracker* tracker;
    std::vector<CloudSlice*> cloudSlices;
    ThreadMutexObject<int> latestPoseId;
    ThreadMutexObject<bool> cloudSliceProcessorFinished;
    std::atomic<bool> trackerFinished;
    ThreadDataPack() : tracker(0), latestPoseId(0), cloudSliceProcessorFinished(false), trackerFinished(false) {}
};

// the backend thread's own context (its own stream): kt_ctx is not shared between threads
static kt_ctx* consumerContext()
{
    static kt_ctx* ctx = 0;
    if (!ctx) ktSafeCall(kt_ctx_create(ConfigArgs::get().gpu, &ctx));
    return ctx;
}

static bool processOnce(ThreadDataPack& threadPack, int& latestPushedCloud, ThreadMutexObject<uint64_t>& lagTime)
{
    std::unique_lock<std::mutex> lock(threadPack.tracker->cloudMutex);
    // the reference waits unconditionally (CloudSliceProcessor.cpp:42); a bounded wait keeps the test from hanging on a lost signal
    threadPack.tracker->cloudSignal.wait_for(lock, std::chrono::milliseconds(50));
    std::vector<CloudSlice*>* trackerSlices = &threadPack.tracker->getCloudSlices();
    const int numClouds = (int)trackerSlices->size();
    const bool cycledMutex = threadPack.tracker->cycledMutex;
    if (cycledMutex) threadPack.tracker->cycledMutex = false;

    if (threadPack.cloudSlices.size() == 0) {
        const uint64_t initTime = threadPack.tracker->init_utime.getValue();
        if (initTime == std::numeric_limits<unsigned long long>::max()) return true;
        kt::Matrix3f lastRotation = threadPack.tracker->getLastRotation();
        kt::Vector3f lastTranslation = threadPack.tracker->getLastTranslation();
        threadPack.cloudSlices.push_back(new CloudSlice(new CloudSlice::PointCloud(), CloudSlice::FIRST, CloudSlice::FAIL, lastTranslation, lastRotation,
                                                        initTime, 0, 0, 0, 0, 0, &threadPack.tracker->placeRecognitionBuffer[0]));
        threadPack.cloudSlices.back()->processedCloud = new CloudSlice::PointCloudNormal();
        threadPack.latestPoseId.assignAndNotifyAll((int)threadPack.cloudSlices.size());
    }
    lock.unlock();

    if (cycledMutex || latestPushedCloud < numClouds) {
        while (latestPushedCloud < numClouds) {
            CloudSlice* s = trackerSlices->at(latestPushedCloud);
            // CloudSliceProcessor.cpp:87-163: weight cull, voxel grid at the voxel leaf size, kNN(20) normals -- one call on the GPU
            s->processedCloud = new CloudSlice::PointCloudNormal(s->cloud->size());
            size_t np = 0;
            if (s->cloud->size()) {
                const float3& vs = Volume::get().getVoxelSizeMeters();
                const float leafSize = std::max(vs.x, std::max(vs.y, vs.z));
                static_assert(sizeof(PointXYZRGBNormal) == sizeof(kt_point_xyzrgbnormal), "processedCloud layout");
                ktSafeCall(kt_slice_process(consumerContext(), s->cloud->data(), s->cloud->size(), ConfigArgs::get().weightCull, leafSize, 20,
                                            reinterpret_cast<kt_point_xyzrgbnormal*>(s->processedCloud->data()), &np));
            }
            s->processedCloud->resize(np);
            threadPack.cloudSlices.push_back(s);
            threadPack.latestPoseId.assignAndNotifyAll((int)threadPack.cloudSlices.size());
            latestPushedCloud++;
        }
    }
    if (latestPushedCloud) lagTime.assignValue(trackerSlices->at(latestPushedCloud - 1)->lagTime);
    if (threadPack.cloudSlices.size() && threadPack.cloudSlices.back()->dimension == CloudSlice::FINAL) {
        threadPack.cloudSliceProcessorFinished.assignAndNotifyAll(true);
        lagTime.assignValue(0);
        return false;
    }
    return true;
}

int main(int argc, char** argv)
{
    const ConfigArgs& args = ConfigArgs::get(argc, argv);
    if (args.logFile.empty()) { ConfigArgs::usage(argv[0]); return 1; }
    Resolution::get(args.width, args.height);
    Volume::get(args.volumeSize, args.volumeResolution);
    const Intr intr(528.0f * args.width / 640.0f, 528.0f * args.height / 480.0f, 320.0f * args.width / 640.0f, 240.0f * args.height / 480.0f);
    RawLogReader log(args.logFile);
    TrackerInterface tracker(&log, intr, false);
    if (args.extractOverlap) tracker.enableOverlap();
    ThreadDataPack pack;
    pack.tracker = tracker.getFrontend();
    pack.tracker->tsdfRequest.assignValue(true);     // the GUI's "draw the live TSDF" switch (PangoVis)
    pack.tracker->liveViewsEnabled = true;

    std::thread consumer([&]() {
        int latestPushedCloud = 0;
        ThreadMutexObject<uint64_t> lagTime(0);
        int idle = 0;
        while (processOnce(pack, latestPushedCloud, lagTime)) {
            if (pack.trackerFinished && ++idle > 200) break;   // 10 s after the tracker has finished without a FINAL slice: give up
        }
    });
    while (tracker.process()) {}
    pack.trackerFinished = true;
    consumer.join();

    KintinuousTracker* fe = pack.tracker;
    int loops = 0;
    for (size_t i = 0; i < fe->densePoseGraph.size(); ++i) loops += fe->densePoseGraph[i].isLoopPose;
    const bool finished = pack.cloudSliceProcessorFinished.getValue();
    std::printf("consumed %zu finished %d first_utime %llu pr %d loops %d poses %zu latest %d live %zu %d first_frame %d\n", pack.cloudSlices.size(), (int)finished,
                (unsigned long long)fe->init_utime.getValue(), fe->placeRecognitionId.getValue(), loops, fe->densePoseGraph.size(),
                fe->latestDensePoseId.getValue(), fe->getLiveTsdf() ? fe->getLiveTsdf()->cloud->size() : (size_t)0, fe->getLiveImage() ? 1 : 0,
                (fe->firstRgbImage.getValue() && fe->firstDepthData.getValue()) ? 1 : 0);
    for (size_t i = 1; i < pack.cloudSlices.size(); ++i)
        std::printf("slice %zu dim %d points %zu processed %zu pr %d\n", i, (int)pack.cloudSlices[i]->dimension, pack.cloudSlices[i]->cloud->size(),
                    pack.cloudSlices[i]->processedCloud->size(),
                    pack.cloudSlices[i]->placeRecognitionFrame ? (int)(pack.cloudSlices[i]->placeRecognitionFrame - fe->placeRecognitionBuffer) : -1);
    delete pack.cloudSlices[0];   // the FIRST slice is the consumer's; the others belong to the tracker
    return finished ? 0 : 2;
}
