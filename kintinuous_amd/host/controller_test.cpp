// controller_test -- MainController (src/MainController.cpp) minus the GUI, against this shell: setup() builds the log reader, the
// TrackerInterface (loadTrajectory with -p) and the CloudSliceProcessor exactly as MainController::setup does (:93-141), mainLoop()
// starts every component with ThreadObject::start on a thread of its own and joins them (:142-170), complete() raises endRequested
// (:235-238) and save() writes the cloud through CloudSliceProcessor::save once the run is finalised (:240-265).  What PangoVis would do
// -- the window, its "limit" check box, its Save / End buttons -- is the few lines of main() below: the throttle is switched off, the
// run is ended after -end <K> frames if asked for, and the cloud is saved at the end.  It compiles against the same public surface the
// unchanged MainController uses (TrackerInterface : ThreadObject, endRequested, loadTrajectory, enableOverlap, getFrontend,
// ThreadDataPack::get().assignFrontend / finalised / *Finished flags, CloudSliceProcessor : ThreadObject, save) -- compiling and
// running to the same .poses / .pcd as kintinuous_hip is the test (tests/test_gpu_host_shell.py).
// Usage: controller_test -l <log.klg> [tracker options] [-end <K>] [-limit]
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <thread>
#include <vector>

#include "CloudSliceProcessor.h"
#include "TrackerInterface.h"

class MainController {
  public:
    MainController(int argc, char* argv[]) : depthIntrinsics(0), trackerInterface(0), cloudSliceProcessor(0), rawRead(0), logRead(0)
    {
        ConfigArgs::get(argc, argv);
        assert(!MainController::controller);
        MainController::controller = this;
    }
    virtual ~MainController() { delete depthIntrinsics; }

    int start() { return setup() ? mainLoop() : -1; }

    // MainController.cpp:93-141
    bool setup()
    {
        Volume::get(ConfigArgs::get().volumeSize, ConfigArgs::get().volumeResolution);
        Stopwatch::get().setCustomSignature(43543534);
        kt::device::context(ConfigArgs::get().gpu);   // cudaSafeCall(cudaSetDevice(ConfigArgs::get().gpu))
        loadCalibration();
        std::cout << "Point resolution: " << ((int)((Volume::get().getVoxelSizeMeters().x * 1000.0f) * 10.0f)) / 10.0f << " millimetres" << std::endl;
        rawRead = new RawLogReader;   // (LiveLogReader -- an OpenNI camera -- is not part of this path)
        logRead = static_cast<LogReader*>(rawRead);
        ThreadDataPack::get();
        trackerInterface = new TrackerInterface(logRead, *depthIntrinsics);
        if (ConfigArgs::get().trajectoryFile.size()) {
            std::cout << "Load trajectory: " << ConfigArgs::get().trajectoryFile << std::endl;
            trackerInterface->loadTrajectory(ConfigArgs::get().trajectoryFile);
        }
        systemComponents.push_back(trackerInterface);
        ThreadDataPack::get().assignFrontend(trackerInterface->getFrontend());
        cloudSliceProcessor = new CloudSliceProcessor();
        systemComponents.push_back(cloudSliceProcessor);
        if (ConfigArgs::get().extractOverlap) trackerInterface->enableOverlap();
        // no MeshGenerator, no Deformation / PlaceRecognition threads on this path (:121-141)
        ThreadDataPack::get().meshGeneratorFinished.assignValue(true);
        ThreadDataPack::get().deformationFinished.assignValue(true);
        ThreadDataPack::get().placeRecognitionFinished.assignValue(true);
        return true;
    }

    // MainController.cpp:142-183 without PangoVis: the components run until they end themselves
    int mainLoop()
    {
        for (unsigned int i = 0; i < systemComponents.size(); i++) threads.push_back(new std::thread(&ThreadObject::start, systemComponents.at(i)));
        return 0;
    }
    void join()
    {
        for (size_t i = 0; i < threads.size(); i++) { threads[i]->join(); delete threads[i]; }
        threads.clear();
    }
    void tearDown()
    {
        for (unsigned int i = 0; i < systemComponents.size(); i++) delete systemComponents.at(i);
        systemComponents.clear();
        delete rawRead;
        rawRead = 0;
    }

    // MainController.cpp:185-233 (ElasticFusion format: one line "fx fy cx cy [w h]"; the OpenCV .xml / .yml format needs OpenCV)
    void loadCalibration()
    {
        const std::string& calFile = ConfigArgs::get().calibrationFile;
        int w = ConfigArgs::get().width, h = ConfigArgs::get().height;
        if (calFile.length() > 0) {
            std::ifstream file(calFile);
            std::string line;
            std::getline(file, line);
            double fx, fy, cx, cy, cw, ch;
            const int n = std::sscanf(line.c_str(), "%lg %lg %lg %lg %lg %lg", &fx, &fy, &cx, &cy, &cw, &ch);
            if (n != 4 && n != 6) { std::fprintf(stderr, "Ooops, your calibration file should contain a single line with [fx fy cx cy] or [fx fy cx cy w h]\n"); std::exit(1); }
            depthIntrinsics = new Intr((float)fx, (float)fy, (float)cx, (float)cy);
            if (n == 6) { w = (int)cw; h = (int)ch; }
        } else {
            depthIntrinsics = new Intr(528.01442863461716f, 528.01442863461716f, 320.0f, 267.0f);
        }
        Resolution::get(w, h);
    }

    // proxy functions for the GUI (MainController.cpp:235-265)
    void complete() { trackerInterface->endRequested.assignValue(true); }
    long long save()
    {
        if (!ThreadDataPack::get().finalised.getValue()) return -1;
        return cloudSliceProcessor->save();   // (the reference runs it on a thread of its own, :244)
    }

    static MainController* controller;
    TrackerInterface* trackerInterface;

  private:
    Intr* depthIntrinsics;
    CloudSliceProcessor* cloudSliceProcessor;
    RawLogReader* rawRead;
    LogReader* logRead;
    std::vector<std::thread*> threads;
    std::vector<ThreadObject*> systemComponents;
};
MainController* MainController::controller = 0;

int main(int argc, char* argv[])
{
    int endAfter = 0;
    bool limit = false, stage = false;
    for (int i = 1; i < argc; ++i) {
        if (i + 1 < argc && std::string(argv[i]) == "-end") endAfter = std::atoi(argv[i + 1]);
        limit = limit || std::string(argv[i]) == "-limit";
        stage = stage || std::string(argv[i]) == "-stage";
    }
    MainController controller(argc, argv);
    if (ConfigArgs::get().logFile.empty()) { ConfigArgs::usage(argv[0]); return 1; }
    if (!controller.setup()) return 1;
    ThreadDataPack& pack = ThreadDataPack::get();
    pack.limit.assignValue(limit);   // PangoVis' "limit" check box (default on in the reference: 30 Hz playback)
    if (stage) pack.tracker->enableSliceStage(ConfigArgs::get().weightCull);   // the per-slice stage on the device, behind the extraction
    const auto t0 = std::chrono::steady_clock::now();
    controller.mainLoop();
    if (endAfter > 0) {   // the GUI's "End" button after K frames
        while (pack.trackerFrame.getValue() < endAfter && !pack.finalised.getValue()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        controller.complete();
    }
    controller.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const bool finished = pack.finalised.getValue() && pack.cloudSliceProcessorFinished.getValue();
    const long long saved = controller.save();   // the GUI's "Save" button
    KintinuousTracker* fe = pack.tracker;
    const kt::Vector3f cam = fe->getCurrentGlobalCamera();
    std::printf("controller frames %d slices %zu consumed %zu saved %lld finished %d last camera %.6f %.6f %.6f  %.1f frames/s\n",
                controller.trackerInterface->getCurrentFrame(), fe->getCloudSlices().size(), pack.cloudSlices.size(), saved, (int)finished, cam(0), cam(1), cam(2),
                controller.trackerInterface->getCurrentFrame() / sec);
    pack.reset();
    controller.tearDown();
    return finished && saved >= 0 ? 0 : 2;
}
