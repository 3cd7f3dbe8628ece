// STAND-IN for the reference's own header (utils/ThreadDataPack.h): in an integration the reference's file is used as it is and this
// one is deleted.  Interface glue for the shell's tests, not product code -- do not grow it.
//
// ThreadDataPack.h -- the singleton the reference's threads talk through (utils/ThreadDataPack.h:34-165), cut to the fields the
// tracking + fusion path and its first consumer read and write: the tracker pointer, the slices taken over by CloudSliceProcessor,
// latestPoseId and the hand-shake flags of the end of a run (pauseCapture, finalised, cloudSliceProcessorFinished), the 30 Hz
// `limit` throttle, trackerFrame.  The mesh / loop-closure / deformation members (incrementalMesh, triangles, pointPool,
// loopClosureConstraints, loopOffset, isamOffset ...) belong to backend threads that are not part of this path; their "finished" flags
// are kept so that a controller can mark those components absent the way MainController::setup does (MainController.cpp:121-141).
#pragma once

#include <assert.h>
#include <stdint.h>
#include <vector>

#include "ConfigArgs.h"
#include "KintinuousTracker.h"
#include "ThreadMutexObject.h"

class ThreadDataPack {
  public:
    static ThreadDataPack& get()
    {
        static ThreadDataPack instance;
        return instance;
    }

    virtual ~ThreadDataPack() {}

    void assignFrontend(KintinuousTracker* frontend)
    {
        tracker = frontend;   // (the reference asserts !tracker, :56: one controller per process; tests re-assign after a reset)
    }

    void reset()
    {
        // only the first item is deleted: it is the initial pose slice created by the CloudSliceProcessor, the rest of the pointers
        // are owned by the KintinuousTracker and dealt with there (:62-70)
        if (cloudSlices.size()) delete cloudSlices.at(0);
        cloudSlices.clear();
        pauseCapture.assignValue(false);
        latestLoopId.assignValue(0);
        latestPoseId.assignValue(0);
        latestMeshId.assignValue(0);
        trackerFinished.assignValue(false);
        cloudSliceProcessorFinished.assignValue(false);
        meshGeneratorFinished.assignValue(false);
        placeRecognitionFinished.assignValue(false);
        deformationFinished.assignValue(false);
        trackerFrame.assignValue(0);
        finalised.assignValue(false);
        limit.assignValue(true);
        lastLoopTime.assignValue(0);
        readyForLoop.assignValue(true);
    }

    void notifyVariables()
    {
        latestLoopId.notifyAll();
        latestMeshId.notifyAll();
        latestPoseId.notifyAll();
    }

    ThreadMutexObject<bool> finalised;
    ThreadMutexObject<bool> limit;

    KintinuousTracker* tracker;
    std::vector<CloudSlice*> cloudSlices;

    ThreadMutexObject<uint64_t> lastLoopTime;
    ThreadMutexObject<bool> readyForLoop;

    ThreadMutexObject<int> latestLoopId;
    ThreadMutexObject<int> latestMeshId;
    ThreadMutexObject<int> latestPoseId;
    ThreadMutexObject<bool> trackerFinished;
    ThreadMutexObject<bool> cloudSliceProcessorFinished;
    ThreadMutexObject<bool> meshGeneratorFinished;
    ThreadMutexObject<bool> placeRecognitionFinished;
    ThreadMutexObject<bool> deformationFinished;
    ThreadMutexObject<int> trackerFrame;
    ThreadMutexObject<bool> pauseCapture;

  private:
    ThreadDataPack() : tracker(0) { reset(); }
};
