// consumer_test -- the reference's backend consumer thread (host/CloudSliceProcessor.h = backend/CloudSliceProcessor.cpp:38-181) run
// against this shell while the tracker thread plays a log.  It touches exactly
// what the reference's consumer touches -- tracker->cloudMutex, cloudSignal.wait(cloudMutex), getCloudSlices(), cycledMutex,
// init_utime, getLastRotation / getLastTranslation, placeRecognitionBuffer[0], CloudSlice's 12-argument constructor, processedCloud,
// lagTime, the FINAL dimension -- so compiling and finishing is the test.  Usage: consumer_test -l <log.klg> [tracker options] [-v x]
// prints: consumed <slices> first_utime <t> pr <samples> loops <n> poses <n> live <tsdf points> <image 0|1>
#include <atomic>
#include <cstdio>
#include <limits>
#include <thread>

#include "CloudSliceProcessor.h"
#include "TrackerInterface.h"

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
    ThreadDataPack& pack = ThreadDataPack::get();
    pack.assignFrontend(tracker.getFrontend());
    CloudSliceProcessor processor;
    pack.tracker->tsdfRequest.assignValue(true);     // the GUI's "draw the live TSDF" switch (PangoVis)
    pack.tracker->liveViewsEnabled = true;

    pack.limit.assignValue(false);   // the GUI's 30 Hz throttle (ThreadDataPack::limit) off: play the log as fast as it tracks
    // MainController::mainLoop (MainController.cpp:142-150): every component runs ThreadObject::start on a thread of its own
    std::thread consumer(&ThreadObject::start, static_cast<ThreadObject*>(&processor));
    std::thread producer(&ThreadObject::start, static_cast<ThreadObject*>(&tracker));
    producer.join();   // ends after finalise() and the hand-shake on cloudSliceProcessorFinished (TrackerInterface.cpp:57-71)
    pack.trackerFinished.assignValue(true);
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
    pack.reset();   // deletes the FIRST slice (the processor's own); the others belong to the tracker
    return finished ? 0 : 2;
}
