// kintinuous_hip -- headless driver of the tracking + fusion path over a .klg log: the part of the reference's
// `Kintinuous -l log.klg [-c calib] [-s size] [-t shift] [-r|-ri] [-fod] [-sm] ...` run (src/Kintinuous.cpp,
// MainController.cpp:73-170) that ends at the CloudSlices and the .poses file.  Extra options: -n <N>, -w/-h, -o <prefix>,
// -ops (compose every frame from the internal.h operators instead of the device-resident tracker), -pcd (run the CloudSliceProcessor thread
// behind the tracker and save <prefix>.pcd the way CloudSliceProcessor::save does), -ppm (write the model views).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>

#include "CloudSliceProcessor.h"
#include "TrackerInterface.h"

static Intr loadCalibration(const std::string& file, int width, int height)
{
    // MainController.cpp:121-150: a calibration file holds "fx fy cx cy"; the default is 528 / 528 / 320 / 240 at VGA
    Intr k(528.0f * width / 640.0f, 528.0f * height / 480.0f, 320.0f * width / 640.0f, 240.0f * height / 480.0f);
    if (file.size()) {
        std::ifstream f(file.c_str());
        double fx, fy, cx, cy;
        if (f >> fx >> fy >> cx >> cy) k = Intr((float)fx, (float)fy, (float)cx, (float)cy);
        else { std::fprintf(stderr, "cannot read calibration %s\n", file.c_str()); std::exit(1); }
    }
    return k;
}

// -pcdraw (debug): every extracted slice as the tracker produced it, appended into one binary PCD in pcl::PointXYZRGB field order
// (x y z rgb) -- the input of the slice processor, for comparing extraction paths point by point
static bool writeRawPcd(const std::string& file, const std::vector<CloudSlice*>& slices)
{
    size_t n = 0;
    for (size_t i = 0; i < slices.size(); ++i) n += slices[i]->cloud->size();
    FILE* f = std::fopen(file.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
                    "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n", n, n);
    for (size_t i = 0; i < slices.size(); ++i)
        for (size_t k = 0; k < slices[i]->cloud->size(); ++k) {
            const unsigned char* p = reinterpret_cast<const unsigned char*>(&(*slices[i]->cloud)[k]);
            std::fwrite(p, 1, 12, f);        // x y z
            std::fwrite(p + 16, 1, 4, f);    // b g r a
        }
    return std::fclose(f) == 0;
}

// -ppm: the reference's live views of the final model, without a window: <prefix>_model.ppm (shaded), <prefix>_color.ppm (fused
// colour), <prefix>_depth.pgm (16-bit millimetres)
static void writeViews(KintinuousTracker* fe, const std::string& prefix)
{
    fe->getImage();
    fe->getModelDepth();
    const int rows = Resolution::get().rows(), cols = Resolution::get().cols();
    std::vector<PixelRGB> img;
    int c;
    const char* names[2] = {"_model.ppm", "_color.ppm"};
    for (int k = 0; k < 2; ++k) {
        (k == 0 ? fe->modelSurface : fe->modelColor).download(img, c);
        FILE* f = std::fopen((prefix + names[k]).c_str(), "wb");
        if (!f) continue;
        std::fprintf(f, "P6\n%d %d\n255\n", cols, rows);
        std::fwrite(img.data(), 3, img.size(), f);
        std::fclose(f);
    }
    FILE* f = std::fopen((prefix + "_depth.pgm").c_str(), "wb");
    if (!f) return;
    std::fprintf(f, "P5\n%d %d\n65535\n", cols, rows);
    for (size_t i = 0; i < fe->modelDepthHost.size(); ++i) {
        const unsigned char be[2] = {(unsigned char)(fe->modelDepthHost[i] >> 8), (unsigned char)(fe->modelDepthHost[i] & 255)};
        std::fwrite(be, 1, 2, f);
    }
    std::fclose(f);
}

// -rank R -world W -comm <file> [-gk K]: one process per GPU, each on its own log; after the last frame the ranks exchange their K most
// recent dense poses (default 1: the final pose) with the path's single collective (kt_pose_gather: one RCCL all-gather over xGMI).
// EVERY rank contributes the same K -- a collective with unequal counts hangs -- so K comes from the command line, not from the length
// of the rank's own log, and a rank whose log gave fewer than K poses fails BEFORE it joins the communicator.  Rank 0 writes the
// 128-byte RCCL id to <file> (atomically, via rename); the other ranks wait for it.  A watchdog ends the process if the rendezvous or
// the gather does not complete (a rank that died leaves the others inside RCCL for ever): KT_COMM_TIMEOUT_S, default 120.
static bool gatherPoses(KintinuousTracker* fe, int rank, int world, const std::string& idFile, int k)
{
    const int have = kt_tracker_num_poses(fe->handle());
    if (k < 1 || have < k) {
        std::fprintf(stderr, "rank %d: %d dense poses, the gather needs %d from every rank (-gk)\n", rank, have, k);
        return false;
    }
    std::atomic<bool> done(false);
    const char* to = std::getenv("KT_COMM_TIMEOUT_S");
    const int timeout_s = to ? std::atoi(to) : 120;
    std::thread watchdog([&done, timeout_s, rank]() {
        for (int waited = 0; waited < timeout_s * 10 && !done; ++waited) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!done) {
            std::fprintf(stderr, "rank %d: pose gather did not complete within %d s (another rank missing?)\n", rank, timeout_s);
            std::_Exit(3);
        }
    });
    struct Join { std::atomic<bool>& d; std::thread& t; ~Join() { d = true; t.join(); } } join{done, watchdog};
    unsigned char id[KT_COMM_ID_BYTES];
    if (rank == 0) {
        ktSafeCall(kt_comm_unique_id(id));
        FILE* f = std::fopen((idFile + ".tmp").c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id) || std::fclose(f) != 0) return false;
        if (std::rename((idFile + ".tmp").c_str(), idFile.c_str()) != 0) return false;
    } else {
        for (;;) {   // (bounded by the watchdog)
            FILE* f = std::fopen(idFile.c_str(), "rb");
            if (f) {
                const size_t got = std::fread(id, 1, sizeof(id), f);
                std::fclose(f);
                if (got == sizeof(id)) break;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    kt_comm* comm = 0;
    ktSafeCall(kt_comm_init(kt::device::context(), rank, world, id, &comm));
    std::vector<float> all((size_t)world * k * 16);
    ktSafeCall(kt_pose_gather(comm, fe->handle(), k, all.data()));
    ktSafeCall(kt_comm_destroy(comm));
    for (int r = 0; r < world; ++r) {
        const float* p = &all[((size_t)r * k + (k - 1)) * 16];
        std::printf("rank %d sees stream %d: last camera %.6f %.6f %.6f (%d poses gathered, %zu bytes)\n", rank, r, p[3], p[7], p[11], k,
                    all.size() * sizeof(float));
    }
    return true;
}

int main(int argc, char** argv)
{
    const ConfigArgs& args = ConfigArgs::get(argc, argv);
    if (args.help || args.logFile.empty()) { ConfigArgs::usage(argv[0]); return args.help ? 0 : 1; }
    bool ops = false, pcd = false, pcdraw = false, ppm = false, noStage = false;
    int rank = 0, world = 0, gatherCount = 1;
    std::string commFile;
    for (int i = 1; i < argc; ++i) {
        ops = ops || std::string(argv[i]) == "-ops";
        pcd = pcd || std::string(argv[i]) == "-pcd";
        pcdraw = pcdraw || std::string(argv[i]) == "-pcdraw";
        noStage = noStage || std::string(argv[i]) == "-nostage";   // debug: the slice processor thread calls kt_slice_process itself
        ppm = ppm || std::string(argv[i]) == "-ppm";
        if (i + 1 < argc && std::string(argv[i]) == "-rank") rank = std::atoi(argv[i + 1]);
        if (i + 1 < argc && std::string(argv[i]) == "-world") world = std::atoi(argv[i + 1]);
        if (i + 1 < argc && std::string(argv[i]) == "-comm") commFile = argv[i + 1];
        if (i + 1 < argc && std::string(argv[i]) == "-gk") gatherCount = std::atoi(argv[i + 1]);
    }

    Resolution::get(args.width, args.height);
    Volume::get(args.volumeSize, args.volumeResolution);
    const Intr intr = loadCalibration(args.calibrationFile, args.width, args.height);

    RawLogReader log(args.logFile);
    TrackerInterface tracker(&log, intr, ops);
    if (args.extractOverlap) tracker.enableOverlap();  // MainController.cpp:187-190

    // -pcd: the backend's first thread runs next to the tracker, as in MainController (CloudSliceProcessor.cpp): it takes every slice
    // at the moment the tracker hands it over and fills its processedCloud on the GPU, on a context and stream of its own
    ThreadDataPack& pack = ThreadDataPack::get();
    pack.assignFrontend(tracker.getFrontend());
    // the slice stage itself runs on the device behind every extraction (the operator path hands raw slices to the processor instead)
    if (pcd && !ops && !noStage) tracker.getFrontend()->enableSliceStage(args.weightCull);
    CloudSliceProcessor sliceProcessor;
    pack.limit.assignValue(false);   // the GUI's 30 Hz throttle (ThreadDataPack::limit) off: play the log as fast as it tracks
    // Components run as in MainController::mainLoop (MainController.cpp:142-150): ThreadObject::start on a thread each.  Without -pcd no
    // slice processor runs; it is then marked finished up front, the way MainController::setup marks absent components (:121-141), so
    // that the tracker's end-of-log hand-shake (TrackerInterface.cpp:66-69) has nothing to wait for.
    std::thread sliceThread;
    if (pcd) sliceThread = std::thread(&ThreadObject::start, static_cast<ThreadObject*>(&sliceProcessor));
    else pack.cloudSliceProcessorFinished.assignValue(true);

    const auto t0 = std::chrono::steady_clock::now();
    std::thread trackerThread(&ThreadObject::start, static_cast<ThreadObject*>(&tracker));
    trackerThread.join();
    const int frames = tracker.getCurrentFrame();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    pack.trackerFinished.assignValue(true);
    if (pcd) sliceThread.join();   // the slices are the processor's until it has ended (it rewrites slice->cloud in place)

    KintinuousTracker* fe = tracker.getFrontend();
    // -pcdraw is the extraction as the tracker produced it: only meaningful without -pcd, whose processor down-samples slice->cloud in place
    if (pcdraw && !pcd && !writeRawPcd(args.saveFile + ".raw.pcd", fe->getCloudSlices())) std::fprintf(stderr, "cannot write %s.raw.pcd\n", args.saveFile.c_str());
    if (pcdraw && pcd) std::fprintf(stderr, "-pcdraw ignored with -pcd (the slice processor rewrites the slices in place)\n");
    size_t points = 0;   // with -pcd: the down-sampled slices (the processor has ended); without: the raw extraction
    for (size_t i = 0; i < fe->getCloudSlices().size(); ++i) points += fe->getCloudSlices()[i]->cloud->size();
    if (pcd && (!pack.cloudSliceProcessorFinished.getValue() || sliceProcessor.save() < 0)) std::fprintf(stderr, "cannot write %s.pcd\n", args.saveFile.c_str());
    if (ppm) writeViews(fe, args.saveFile);
    const kt::Vector3f cam = fe->getCurrentGlobalCamera();
    std::printf("frames %d  slices %zu  points %zu  last camera %.6f %.6f %.6f  %.1f frames/s (incl. file I/O and uploads)  path %s\n", frames,
                fe->getCloudSlices().size(), points, cam(0), cam(1), cam(2), frames / sec, ops ? "operators" : "device-resident");
    if (world > 0 && !ops) {
        if (commFile.empty() || !gatherPoses(fe, rank, world, commFile, gatherCount)) { std::fprintf(stderr, "pose gather failed (-comm <file> shared by all ranks)\n"); return 1; }
    }
    return 0;
}
