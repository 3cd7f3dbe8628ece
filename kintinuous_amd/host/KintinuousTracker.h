// KintinuousTracker.h -- the per-frame tracking + fusion front end with the reference's public surface
// (frontend/KintinuousTracker.h:82-276, KintinuousTracker.cpp:71-1050).  Two execution paths with identical results:
//   * device-resident (default): one kt_tracker_process_frame call; the Gauss-Newton iterations, shift clears and the
//     predicted-map pyramid stay on the GPU, one host sync per frame;
//   * operator path (operatorPath = true): the frame is composed on the host from the internal.h operators exactly as the
//     reference composes it (bilateralFilter, pyrDown, createVMap, ..., ICPOdometry, integrateTsdfVolume, raycast,
//     resizeVMap) -- the drop-in granularity of the reference, one sync per operator.  All three odometry providers
//     (ICPOdometry, RGBDOdometry for -r / -ri, GroundTruthOdometry for -p), selected as KintinuousTracker.cpp:128-178 does.
// getImage / getModelDepth (KintinuousTracker.cpp:960-981) render the predicted maps without OpenGL; -p ground truth is handled by
// the device-resident path.  The fields other threads of the reference touch (CloudSliceProcessor.cpp:38-83, PlaceRecognition, the GUI)
// are here with their names and protocols: cloudMutex / cloudSignal / cycledMutex around sharedCloudSlices, init_utime,
// firstRgbImage / firstDepthData, placeRecognitionId / placeRecognitionBuffer, latestDensePoseId, tsdfRequest / getLiveTsdf,
// imageAvailable / getLiveImage (boost::mutex -> std::mutex, boost::condition_variable_any -> std::condition_variable_any).
#pragma once

#include <climits>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "CloudSlice.h"
#include "ConfigArgs.h"
#include "GroundTruthOdometry.h"
#include "ICPOdometry.h"
#include "RGBDOdometry.h"
#include "PlaceRecognitionInput.h"
#include "Resolution.h"
#include "ThreadMutexObject.h"
#include "TSDFVolume.h"
#include "Volume.h"

class KintinuousTracker {
  public:
    // KintinuousTracker.h:87-97: what the GUI and the backend threads synchronise on
    ThreadMutexObject<bool> tsdfRequest;
    bool tsdfAvailable;
    std::mutex tsdfMutex;
    bool imageAvailable;
    std::mutex imageMutex;
    bool cycledMutex;
    std::mutex cloudMutex;
    std::condition_variable_any cloudSignal;
    // :135-147
    ThreadMutexObject<uint64_t> init_utime;
    ThreadMutexObject<unsigned char*> firstRgbImage;
    ThreadMutexObject<unsigned short*> firstDepthData;
    unsigned char* lastRgbImage;
    unsigned short* lastDepthData;
    ThreadMutexObject<int> placeRecognitionId;
    static const int PR_BUFFER_SIZE = 3000;
    PlaceRecognitionInput* placeRecognitionBuffer;   // [PR_BUFFER_SIZE] (an array member in the reference; on the heap here)
    ThreadMutexObject<int> latestDensePoseId;

    class DensePose {  // KintinuousTracker.h:151-169
      public:
        DensePose(uint64_t timestamp, const kt::Matrix4f& pose, bool isLoopPose) : timestamp(timestamp), pose(pose), isLoopPose(isLoopPose) {}
        DensePose() : timestamp(0), isLoopPose(false) {}
        uint64_t timestamp;
        kt::Matrix4f pose;
        bool isLoopPose;
    };

    std::vector<DensePose> densePoseGraph;
    CloudSlice::Odometry lastOdometry;

    // depthIntrinsics: fx, fy, cx, cy of the depth camera (the reference passes the 3x3 K as a cv::Mat)
    explicit KintinuousTracker(const Intr& depthIntrinsics, bool operatorPath = false)
        : tsdfRequest(false), tsdfAvailable(false), imageAvailable(false), cycledMutex(false), lastRgbImage(0), lastDepthData(0),
          placeRecognitionId(0), placeRecognitionBuffer(new PlaceRecognitionInput[PR_BUFFER_SIZE]), latestDensePoseId(0),
          lastOdometry(CloudSlice::ICP), intr(depthIntrinsics), operatorPath(operatorPath), fast(0), tsdf_volume_(0), color_volume_(0),
          icp(0), rgbd(0), groundTruth(0), odometryProvider(0), odom_utime(0), overlap(0), parked(false), global_time_(0), current_utime(0), nextSlice(0), nextPrSample(0), liveTsdf(0), liveImage(0), lagTime(0)
    {
        init_utime.assignValue(std::numeric_limits<unsigned long long>::max());   // KintinuousTracker.cpp:124
        const ConfigArgs& args = ConfigArgs::get();
        N = Volume::get().getResolution();
        if (!operatorPath) {
            kt_tracker_config cfg = {};
            cfg.cols = Resolution::get().cols();
            cfg.rows = Resolution::get().rows();
            cfg.N = N;
            cfg.fx = intr.fx; cfg.fy = intr.fy; cfg.cx = intr.cx; cfg.cy = intr.cy;
            cfg.volume_size = Volume::get().getVolumeSize();
            cfg.voxel_shift = args.voxelShift;
            cfg.overlap = 0;
            cfg.static_mode = args.staticMode;
            cfg.dynamic_cube = args.dynamicCube;
            cfg.use_rgbd = args.useRGBD;
            cfg.use_rgbd_icp = args.useRGBDICP;
            cfg.fast_odometry = args.fastOdometry;
            cfg.disable_color_angle = args.disableColorAngleWeight;
            cfg.place_recognition = args.vocabFile.size() ? 1 : 0;
            config = cfg;
            lastOdometry = (args.useRGBD || args.useRGBDICP) ? CloudSlice::RGBD : CloudSlice::ICP;
            if (args.trajectoryFile.size()) {  // KintinuousTracker.cpp:128-138: ground truth wins over the odometry flags
                loadTrajectory(args.trajectoryFile);
                lastOdometry = CloudSlice::GROUNDTRUTH;
            }
            return;
        }
        // KintinuousTracker.cpp:86-118
        const float vs = Volume::get().getVolumeSize();
        volumeBasis = kt::Vector3f(vs * 0.5f, vs * 0.5f, vs * 0.5f);
        if (args.staticMode || args.dynamicCube) volumeBasis(2) = vs * 0.5f - (float)(((double)vs * 0.5) + (args.staticMode ? 0.45 : 0));
        tsdf_volume_ = new TsdfVolume(N);
        tsdf_volume_->setSize(kt::Vector3f(vs, vs, vs));
        tsdf_volume_->setTsdfTruncDist(std::max(0.01f, vs / 100.0f));
        color_volume_ = new ColorVolume(*tsdf_volume_);
        cloud_device_.create((size_t)Resolution::get().numPixels() * 3);
        allocateBuffers();
        // the odometry provider, KintinuousTracker.cpp:128-178: a trajectory file wins over the odometry flags
        if (args.trajectoryFile.size()) {
            loadTrajectory(args.trajectoryFile);
            groundTruth = new GroundTruthOdometry(tvecs_, rmats_, camera_trajectory, odom_utime);
            odometryProvider = groundTruth;
            lastOdometry = CloudSlice::GROUNDTRUTH;
        } else if (args.useRGBD || args.useRGBDICP) {
            rgbd = new RGBDOdometry(tvecs_, rmats_, vmaps_g_prev_, nmaps_g_prev_, vmaps_curr_, nmaps_curr_, intr);
            odometryProvider = rgbd;
            lastOdometry = CloudSlice::RGBD;
        } else {
            icp = new ICPOdometry(tvecs_, rmats_, vmaps_g_prev_, nmaps_g_prev_, vmaps_curr_, nmaps_curr_, intr, args.fastOdometry);
            odometryProvider = icp;
        }
        reset();
    }

    virtual ~KintinuousTracker()
    {
        for (size_t i = 0; i < sharedCloudSlices.size(); ++i) delete sharedCloudSlices[i];
        if (fast) kt_tracker_destroy(fast);
        delete liveTsdf;
        delete liveImage;
        delete[] firstRgbImage.getValue();
        delete[] firstDepthData.getValue();
        delete[] placeRecognitionBuffer;
        delete icp;
        delete rgbd;
        delete groundTruth;
        delete color_volume_;
        delete tsdf_volume_;
    }

    void processFrame(const DeviceArray2D<unsigned short>& depth, const DeviceArray2D<PixelRGB>& colors, unsigned char* rgbImage,
                      unsigned short* depthData, uint64_t timestamp, bool compression = false, uint8_t* lastCompressedDepth = 0,
                      int depthSize = 0, uint8_t* lastCompressedImage = 0, int imageSize = 0)
    {
        lagTime = nowMicros();
        lastRgbImage = rgbImage;
        lastDepthData = depthData;
        current_utime = timestamp;
        frameCompression = compression; frameCompressedDepth = lastCompressedDepth; frameDepthSize = depthSize;
        frameCompressedImage = lastCompressedImage; frameImageSize = imageSize;
        if (!operatorPath) {
            ensureFast();
            const int before = global_time_;
            ktSafeCall(kt_tracker_process_frame(fast, depth.ptr(), reinterpret_cast<const uint8_t*>(colors.ptr()), timestamp));
            syncFromFast();
            if (global_time_ == before) return;  // no trajectory entry for this timestamp: the frame was dropped (:460-463)
        } else {
            const int before = global_time_;
            processFrameOperators(depth, colors, timestamp);
            if (global_time_ == before) return;  // dropped: no trajectory entry for this timestamp (:460-463)
        }
        if (global_time_ > 1 && ConfigArgs::get().saveFile.size()) outputPose(timestamp, lastRotation);
    }

    // Host-resident frames with read-ahead (device-resident path only): announceFrame() stages a frame that a LATER processFrameHost()
    // call with the same two pointers will consume -- its upload and pose-independent stages overlap the frames before it.
    void announceFrame(const unsigned short* depthData, const unsigned char* rgbImage)
    {
        if (operatorPath) return;
        ensureFast();
        ktSafeCall(kt_tracker_prefetch_frame_host(fast, depthData, rgbImage));
    }
    void processFrameHost(unsigned short* depthData, unsigned char* rgbImage, uint64_t timestamp, bool compression = false,
                          uint8_t* lastCompressedDepth = 0, int depthSize = 0, uint8_t* lastCompressedImage = 0, int imageSize = 0)
    {
        lagTime = nowMicros();
        lastRgbImage = rgbImage;
        lastDepthData = depthData;
        current_utime = timestamp;
        frameCompression = compression; frameCompressedDepth = lastCompressedDepth; frameDepthSize = depthSize;
        frameCompressedImage = lastCompressedImage; frameImageSize = imageSize;
        ensureFast();
        const int before = global_time_;
        ktSafeCall(kt_tracker_process_frame_host(fast, depthData, rgbImage, timestamp));
        syncFromFast();
        if (global_time_ == before) return;  // dropped by the ground-truth trajectory lookup (:460-463)
        if (global_time_ > 1 && ConfigArgs::get().saveFile.size()) outputPose(timestamp, lastRotation);
    }

    // KintinuousTracker.cpp:960-969: shaded + colour view of the current predicted map into modelSurface / modelColor
    void getImage()
    {
        LightSource light;
        light.number = 1;
        const float vs = Volume::get().getVolumeSize();
        light.pos[0].x = vs * (-3.f); light.pos[0].y = vs * (-3.f); light.pos[0].z = vs * (-3.f);
        if (operatorPath) { generateImage(vmaps_g_prev_[0], nmaps_g_prev_[0], vmap_curr_color, light, modelSurface, modelColor); return; }
        ensureFast();
        const int rows = Resolution::get().rows(), cols = Resolution::get().cols();
        modelSurface.create(rows, cols);
        modelColor.create(rows, cols);
        ktSafeCall(kt_generate_image(kt::device::context(), kt_tracker_vmap_g_prev(fast, 0), kt_tracker_nmap_g_prev(fast, 0),
                                     kt_tracker_vmap_curr_color(fast), cols, rows, kt::abi(light.pos[0]), light.number, &modelSurface.ptr()->r,
                                     &modelColor.ptr()->r));
    }
    // KintinuousTracker.cpp:971-981: depth (mm) of the predicted map from the current pose, into modelDepth / modelDepthHost
    void getModelDepth()
    {
        kt::Matrix3f Rinv;
        kt_host_mat33_inverse(lastRotation.data(), Rinv.data());
        const int rows = Resolution::get().rows(), cols = Resolution::get().cols();
        modelDepth.create(rows, cols);
        const float* v = operatorPath ? vmaps_g_prev_[0].ptr() : (ensureFast(), kt_tracker_vmap_g_prev(fast, 0));
        const float* n = operatorPath ? nmaps_g_prev_[0].ptr() : kt_tracker_nmap_g_prev(fast, 0);
        ktSafeCall(kt_generate_depth(kt::device::context(), reinterpret_cast<const kt_mat33*>(Rinv.data()), lastTranslation.data(), v, n, cols, rows,
                                     modelDepth.ptr()));
        int c;
        modelDepth.download(modelDepthHost, c);
    }
    DeviceArray2D<PixelRGB> modelSurface, modelColor;   // KintinuousTracker.h:231-235
    DeviceArray2D<unsigned short> modelDepth;
    std::vector<unsigned short> modelDepthHost;

    kt::Vector3f getVolumeOffset() const { return volumeBasisValue(); }
    void setParked(const bool park)
    {
        parked = park;
        if (fast) ktSafeCall(kt_tracker_set_parked(fast, park));
    }
    // tvecs_.back() - volumeBasis (KintinuousTracker.cpp:993-996)
    kt::Vector3f getLastTranslation() const
    {
        const kt::Vector3f b = volumeBasisValue();
        return kt::Vector3f(lastTranslation(0) - b(0), lastTranslation(1) - b(1), lastTranslation(2) - b(2));
    }
    kt::Vector3f getVoxelSize() const
    {
        const float v = Volume::get().getVolumeSize() / (float)N;
        return kt::Vector3f(v, v, v);
    }
    kt::Matrix3f getLastRotation() const { return lastRotation; }
    kt::Vector3f getCurrentGlobalCamera() const { return currentGlobalCamera; }
    std::vector<CloudSlice*>& getCloudSlices() { return sharedCloudSlices; }
    CloudSlice* getLiveTsdf() { return liveTsdf; }     // KintinuousTracker.cpp:1065-1073
    CloudSlice* getLiveImage() { return liveImage; }
    // The backend's CloudSliceProcessor stage (weight cull, voxel grid, 20-NN normals; backend/CloudSliceProcessor.cpp:87-163) on the
    // device, right behind every slab extraction: slices then reach getCloudSlices() with processedCloud filled.  Device-resident path
    // only (the operator path hands raw slices over; host/CloudSliceProcessor.h processes those itself).
    void enableSliceStage(int weightCull)
    {
        sliceStage = true;
        sliceStageCull = weightCull;
        if (fast) ktSafeCall(kt_tracker_enable_slice_stage(fast, 1, weightCull, 20));
    }
    void setOverlap(int o)
    {
        overlap = o;
        config.overlap = o;
        if (fast) { kt_tracker_destroy(fast); fast = 0; }
    }

    // KintinuousTracker::loadTrajectory (KintinuousTracker.cpp:216-260): lines "utime,x,y,z,qx,qy,qz,qw"; the poses themselves are
    // built inside the library (kt_tracker_load_trajectory)
    void loadTrajectory(const std::string& filename)
    {
        FILE* f = std::fopen(filename.c_str(), "r");
        if (!f) {
            std::fprintf(stderr, "cannot open trajectory file %s\n", filename.c_str());
            std::exit(1);
        }
        haveTrajectory = true;
        // the reference fills a map keyed by utime (camera_trajectory[utime] = T, :253): loading a file again -- the constructor does it
        // for -p, MainController::setup does it once more (MainController.cpp:112-116) -- replaces the entries.  So does this.
        trajectoryTimes.clear();
        trajectoryPoses.clear();
        char line[512];
        double trajSum = 0.0;
        bool first = true;
        float lastT[3] = {0, 0, 0};
        while (std::fgets(line, sizeof(line), f)) {
            unsigned long long utime;
            float v[7];
            if (std::sscanf(line, "%llu,%f,%f,%f,%f,%f,%f,%f", &utime, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]) != 8) continue;
            if (!first) {
                const float d[3] = {v[0] - lastT[0], v[1] - lastT[1], v[2] - lastT[2]};
                trajSum += std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            }
            first = false;
            for (int k = 0; k < 3; ++k) lastT[k] = v[k];
            trajectoryTimes.push_back(utime);
            trajectoryPoses.insert(trajectoryPoses.end(), v, v + 7);
        }
        std::fclose(f);
        std::printf("Done loading ground truth, length: %g\n", trajSum);
        if (operatorPath) {
            for (size_t k = 0; k < trajectoryTimes.size(); ++k)
                kt_host_trajectory_pose(&trajectoryPoses[k * 7], camera_trajectory[trajectoryTimes[k]].data());
        } else if (fast) {
            ktSafeCall(kt_tracker_load_trajectory(fast, (int)trajectoryTimes.size(), trajectoryTimes.data(), trajectoryPoses.data()));
        }
    }


    void finalise()
    {
        if (!operatorPath) {
            ensureFast();
            ktSafeCall(kt_tracker_finalise(fast));
            syncFromFast();
            return;
        }
        vWrapCopyUpdate();
        DeviceArray<PointXYZRGB> cloud = tsdf_volume_->fetchCloud(cloud_device_, vWrapCopy, color_volume_->data(), 0, N, 0, N, 0, N, voxelWrap);
        PlaceRecognitionInput* pr = 0;
        if (ConfigArgs::get().vocabFile.size()) {   // :1038-1045: the final slice takes a raw sample of the last frame with it
            lastPlaceRecognitionRot = rmats_.back();
            lastPlaceRecognitionTrans = currentGlobalCamera;
            pr = addToPlaceRecognition(current_utime, lastPlaceRecognitionTrans, lastPlaceRecognitionRot, false);
        }
        pushSlice(cloud, CloudSlice::FINAL, lastRgbImage, lastDepthData, pr);
    }

    void reset()
    {
        global_time_ = 0;
        densePoseGraph.clear();
        // KintinuousTracker.cpp:293-309
        for (int i = 0; i < PR_BUFFER_SIZE; ++i) placeRecognitionBuffer[i].dump();
        placeRecognitionId.assignValue(0);
        latestDensePoseId.assignValue(0);
        nextPrSample = 0;
        delete[] firstRgbImage.getValue();
        delete[] firstDepthData.getValue();
        firstRgbImage.assignValue(0);
        firstDepthData.assignValue(0);
        init_utime.assignValue(std::numeric_limits<unsigned long long>::max());
        for (size_t i = 0; i < sharedCloudSlices.size(); ++i) delete sharedCloudSlices[i];
        sharedCloudSlices.clear();
        nextSlice = 0;
        if (!operatorPath) {
            if (fast) ktSafeCall(kt_tracker_reset(fast));
            return;
        }
        // KintinuousTracker.cpp:262-354
        odom_utime = 0;   // :259
        if (odometryProvider) odometryProvider->reset();   // :307
        rmats_.clear();
        tvecs_.clear();
        rmats_.push_back(kt::Matrix3f());
        tvecs_.push_back(volumeBasis);
        voxelWrap = make_int3(0, 0, 0);
        vWrapCopy = voxelWrap;
        computeGlobalCamera(0);
        lastRotation = rmats_.back();
        lastTranslation = tvecs_.back();
        lastPlaceRecognitionTrans = currentGlobalCamera;   // :290-291
        lastPlaceRecognitionRot = rmats_.back();
        parked = ConfigArgs::get().staticMode;
        tsdf_volume_->reset();
        color_volume_->reset();
        if (ConfigArgs::get().saveFile.size()) {
            FILE* f = std::fopen((ConfigArgs::get().saveFile + ".poses").c_str(), "w");
            if (f) std::fclose(f);
        }
    }

    kt_tracker* handle() { ensureFast(); return fast; }
    // the reference renders the live image for its GUI on every frame the GUI has consumed the previous one; headless callers leave it off
    bool liveViewsEnabled = false;

  private:
    Intr intr;
    bool operatorPath;
    int N;
    // device-resident path
    kt_tracker* fast;
    kt_tracker_config config;
    // operator path state (KintinuousTracker.h:177-250)
    TsdfVolume* tsdf_volume_;
    ColorVolume* color_volume_;
    ICPOdometry* icp;
    RGBDOdometry* rgbd;
    GroundTruthOdometry* groundTruth;
    OdometryProvider* odometryProvider;
    GroundTruthOdometry::Trajectory camera_trajectory;
    uint64_t odom_utime;   // the reference's current_utime as the providers see it: the PREVIOUS frame's stamp during the odometry (:574-575)
    kt::Vector3f volumeBasis;
    std::vector<DeviceArray2D<unsigned short> > depths_curr_;
    std::vector<DeviceArray2D<float> > vmaps_g_prev_, nmaps_g_prev_, vmaps_curr_, nmaps_curr_;
    DeviceArray2D<uchar4> vmap_curr_color;
    DeviceArray2D<float> depthRawScaled_;
    std::vector<kt::Matrix3f> rmats_;
    std::vector<kt::Vector3f> tvecs_;
    int3 voxelWrap, vWrapCopy;
    DeviceArray<PointXYZRGB> cloud_device_;
    // shared
    std::vector<CloudSlice*> sharedCloudSlices;
    int overlap;
    bool parked;
    int global_time_;
    uint64_t current_utime;
    int nextSlice;
    bool haveTrajectory = false;
    bool sliceStage = false;
    int sliceStageCull = 0;
    std::vector<uint64_t> trajectoryTimes;   // -p file, flattened for kt_tracker_load_trajectory
    std::vector<float> trajectoryPoses;
    kt::Matrix3f lastRotation;
    kt::Vector3f lastTranslation, currentGlobalCamera;
    int nextPrSample;                 // samples of the library's place-recognition tap already copied into placeRecognitionBuffer
    CloudSlice* liveTsdf;             // KintinuousTracker.h:243-244
    CloudSlice* liveImage;
    uint64_t lagTime;
    // the operator path's place-recognition tap (KintinuousTracker.cpp:601-624, 706-717, 1038-1045); the device-resident path keeps this
    // state inside kt_tracker
    kt::Matrix3f lastPlaceRecognitionRot;
    kt::Vector3f lastPlaceRecognitionTrans;
    bool frameCompression = false;    // the compressed payloads of the frame being processed, as the log delivered them (:917-958)
    uint8_t* frameCompressedDepth = 0; int frameDepthSize = 0;
    uint8_t* frameCompressedImage = 0; int frameImageSize = 0;
    std::vector<PixelRGB> modelHost;

    static uint64_t nowMicros()   // Stopwatch::getCurrentSystemTime()
    {
        return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    }

    // KintinuousTracker::addToPlaceRecognition :917-958 for sample `id` of the library's tap: the bytes of the frame being processed
    // (compressed as the log had them, or raw) are copied into the next slot of placeRecognitionBuffer
    PlaceRecognitionInput* addToPlaceRecognition(uint64_t utime, const kt::Vector3f& trans, const kt::Matrix3f& rot, bool compression)
    {
        const int nextSlot = placeRecognitionId.getValue();
        if (nextSlot >= PR_BUFFER_SIZE) { std::fprintf(stderr, "placeRecognitionBuffer is full\n"); std::exit(1); }
        const int depthDataSize = compression ? frameDepthSize : Resolution::get().numPixels() * 2;
        const int rgbDataSize = compression ? frameImageSize : Resolution::get().numPixels() * 3;
        unsigned char* depthPr = new unsigned char[depthDataSize];
        unsigned char* imgPr = new unsigned char[rgbDataSize];
        std::memcpy(depthPr, compression ? (const void*)frameCompressedDepth : (const void*)lastDepthData, depthDataSize);
        std::memcpy(imgPr, compression ? (const void*)frameCompressedImage : (const void*)lastRgbImage, rgbDataSize);
        PlaceRecognitionInput& slot = placeRecognitionBuffer[nextSlot];
        slot.dump();
        slot.rgbImage = imgPr;
        slot.imageSize = rgbDataSize;
        slot.depthMap = (unsigned short*)depthPr;
        slot.depthSize = depthDataSize;
        slot.isCompressed = compression;
        slot.originallyCompressed = compression;
        slot.imageIsRaw = !compression;
        slot.utime = utime;
        slot.lagTime = lagTime;
        slot.trans = trans;
        slot.rotation = rot;
        placeRecognitionId++;
        return &slot;
    }

    // mutexOutLiveTsdf / mutexOutLiveImage (KintinuousTracker.cpp:1087-1154, called from processFrame :835-862 when the GUI asked):
    // the whole surface at subsample 1 and the shaded / colour views of the predicted map.  On the device-resident path they are
    // taken after the frame has been fused (the reference takes them just before fusing it: one frame earlier).
    void serveLiveViews()
    {
        if (tsdfRequest.getValue()) {
            bool needed;
            { std::lock_guard<std::mutex> l(tsdfMutex); needed = !tsdfAvailable; }
            if (needed) {
                CloudSlice::PointCloud* pts = new CloudSlice::PointCloud();
                if (operatorPath) {
                    vWrapCopyUpdate();
                    DeviceArray<PointXYZRGB> c = tsdf_volume_->fetchCloud(cloud_device_, vWrapCopy, color_volume_->data(), 0, N, 0, N, 0, N, voxelWrap, 1);
                    c.download(*pts);
                } else {
                    int w[3], wc[3];
                    ktSafeCall(kt_tracker_get_voxel_wrap(fast, w));
                    for (int k = 0; k < 3; ++k) wc[k] = w[k] < 0 ? N - ((-w[k]) % N) : w[k];
                    DeviceArray<PointXYZRGB> buf((size_t)Resolution::get().numPixels() * 3);
                    const float vs = Volume::get().getVolumeSize(), size[3] = {vs, vs, vs};
                    size_t n = 0;
                    ktSafeCall(kt_extract_cloud_slice(kt::device::context(), kt_tracker_volume(fast), size, buf.ptr(), buf.size(), wc, kt_tracker_color_volume(fast),
                                                      0, N, 0, N, 0, N, 1, w, N, &n));
                    pts->resize(n);
                    if (n) ktSafeCall(kt_download(kt::device::context(), pts->data(), buf.ptr(), n * sizeof(PointXYZRGB)));
                }
                std::lock_guard<std::mutex> l(tsdfMutex);
                tsdfAvailable = true;
                delete liveTsdf;
                liveTsdf = new CloudSlice(pts, CloudSlice::TSDF, lastOdometry, currentGlobalCamera, lastRotation, current_utime, nowMicros(), 0);
            }
        }
        bool imageNeeded;
        { std::lock_guard<std::mutex> l(imageMutex); imageNeeded = !imageAvailable && liveViewsEnabled; }
        if (imageNeeded) {
            int cols;
            getImage();
            const size_t bytes = (size_t)Resolution::get().numPixels() * 3;
            modelColor.download(modelHost, cols);
            unsigned char* tsdfImageColor = new unsigned char[bytes];
            std::memcpy(tsdfImageColor, &modelHost[0], bytes);
            modelSurface.download(modelHost, cols);
            unsigned char* tsdfImage = new unsigned char[bytes];
            std::memcpy(tsdfImage, &modelHost[0], bytes);
            std::lock_guard<std::mutex> l(imageMutex);
            imageAvailable = true;
            delete liveImage;
            liveImage = new CloudSlice(0, CloudSlice::TSDF, lastOdometry, currentGlobalCamera, lastRotation, current_utime, nowMicros(), lastRgbImage,
                                       tsdfImageColor, tsdfImage, lastDepthData);
        }
    }

    kt::Vector3f volumeBasisValue() const
    {
        if (operatorPath) return volumeBasis;
        kt::Vector3f b;
        if (fast) { ktSafeCall(kt_tracker_get_volume_basis(fast, b.data())); return b; }
        const float vs = Volume::get().getVolumeSize();
        b = kt::Vector3f(vs * 0.5f, vs * 0.5f, vs * 0.5f);
        if (ConfigArgs::get().staticMode || ConfigArgs::get().dynamicCube)
            b(2) = vs * 0.5f - (float)(((double)vs * 0.5) + (ConfigArgs::get().staticMode ? 0.45 : 0));
        return b;
    }

    void ensureFast()
    {
        if (fast) return;
        ktSafeCall(kt_tracker_create(kt::device::context(), &config, &fast));
        if (parked) ktSafeCall(kt_tracker_set_parked(fast, 1));
        if (sliceStage) ktSafeCall(kt_tracker_enable_slice_stage(fast, 1, sliceStageCull, 20));
        if (haveTrajectory)
            ktSafeCall(kt_tracker_load_trajectory(fast, (int)trajectoryTimes.size(), trajectoryTimes.data(), trajectoryPoses.data()));
        if (ConfigArgs::get().saveFile.size()) {
            FILE* f = std::fopen((ConfigArgs::get().saveFile + ".poses").c_str(), "w");
            if (f) std::fclose(f);
        }
    }

    // pull pose, dense pose graph, place-recognition samples and new slices out of the device-resident tracker, publishing them with
    // the reference's protocol (first frame :523-556, shift slices :1156-1208, dense poses :903-909)
    void syncFromFast()
    {
        ktSafeCall(kt_tracker_get_pose(fast, lastRotation.data(), lastTranslation.data(), currentGlobalCamera.data()));
        const int before = global_time_;
        global_time_ = kt::count(kt_tracker_num_poses(fast));
        for (int i = (int)densePoseGraph.size(); i < global_time_; ++i) {
            uint64_t ts;
            kt::Matrix4f pose;
            int loop;
            ktSafeCall(kt_tracker_get_dense_pose(fast, i, &ts, pose.m, &loop));
            densePoseGraph.push_back(DensePose(ts, pose, loop != 0));
            latestDensePoseId++;
        }
        // the frames the library sampled for place recognition belong to the frame just processed: copy its bytes now
        std::vector<PlaceRecognitionInput*> sampleSlot;
        const int np = kt::count(kt_tracker_num_pr_samples(fast));
        for (; nextPrSample < np; ++nextPrSample) {
            uint64_t ut;
            kt::Vector3f tr;
            kt::Matrix3f ro;
            int poseIndex;
            ktSafeCall(kt_tracker_pr_sample(fast, nextPrSample, &ut, tr.data(), ro.data(), &poseIndex));
            // the final slice's sample is stored uncompressed (:1041)
            const bool finalSample = kt_tracker_num_slices(fast) > 0 && lastSliceIsFinal();
            addToPlaceRecognition(ut, tr, ro, finalSample ? false : frameCompression);
        }
        if (before == 0 && global_time_ >= 1) {   // first frame :523-556
            init_utime.assignValue(current_utime);
            const int n = Resolution::get().numPixels();
            if (lastDepthData && lastRgbImage) {
                unsigned short* firstDepth = new unsigned short[n];
                std::memcpy(firstDepth, lastDepthData, (size_t)n * 2);
                unsigned char* firstImg = new unsigned char[n * 3];
                std::memcpy(firstImg, lastRgbImage, (size_t)n * 3);
                firstDepthData.assignValue(firstDepth);
                firstRgbImage.assignValue(firstImg);
            }
            std::lock_guard<std::mutex> lock(cloudMutex);
            cloudSignal.notify_all();
        }
        const int ns = kt::count(kt_tracker_num_slices(fast));
        for (; nextSlice < ns; ++nextSlice) {
            size_t n;
            int dim, prId;
            ktSafeCall(kt_tracker_slice_info(fast, nextSlice, &n, &dim));
            ktSafeCall(kt_tracker_slice_pr_id(fast, nextSlice, &prId));
            CloudSlice::PointCloud* cloud = new CloudSlice::PointCloud(n);
            if (n) ktSafeCall(kt_tracker_slice_points(fast, nextSlice, cloud->data()));
            kt::Matrix3f R;
            kt::Vector3f cam;
            uint64_t ts;
            ktSafeCall(kt_tracker_slice_pose(fast, nextSlice, R.data(), cam.data(), &ts));
            const bool fin = dim == CloudSlice::FINAL;
            PlaceRecognitionInput* pr = (prId >= 0 && prId < PR_BUFFER_SIZE) ? &placeRecognitionBuffer[prId] : 0;
            // with the slice stage on the device (enableSliceStage) the slice arrives processed: cull, voxel grid and normals ran on the
            // tracker's slice stream right behind the extraction kernel
            long long np = -1;
            ktSafeCall(kt_tracker_slice_processed_info(fast, nextSlice, &np));
            CloudSlice::PointCloudNormal* processed = 0;
            if (np >= 0) {
                processed = new CloudSlice::PointCloudNormal((size_t)np);
                if (np) ktSafeCall(kt_tracker_slice_processed(fast, nextSlice, reinterpret_cast<kt_point_xyzrgbnormal*>(processed->data())));
            }
            std::lock_guard<std::mutex> lock(cloudMutex);
            cycledMutex = true;
            sharedCloudSlices.push_back(new CloudSlice(cloud, (CloudSlice::Dimension)dim, lastOdometry, cam, R, ts, fin ? nowMicros() : lagTime,
                                                       fin ? lastRgbImage : 0, 0, 0, fin ? lastDepthData : 0, pr));
            sharedCloudSlices.back()->processedCloud = processed;
            cloudSignal.notify_all();
        }
        if (global_time_ > before && global_time_ > 1) serveLiveViews();
    }
    bool lastSliceIsFinal()
    {
        size_t n;
        int dim;
        ktSafeCall(kt_tracker_slice_info(fast, kt_tracker_num_slices(fast) - 1, &n, &dim));
        return dim == CloudSlice::FINAL;
    }

    // <saveFile>.poses, KintinuousTracker.cpp:199-218
    void outputPose(uint64_t timestamp, const kt::Matrix3f& Rcurr)
    {
        FILE* f = std::fopen((ConfigArgs::get().saveFile + ".poses").c_str(), "a");
        if (!f) return;
        const kt::Quaternionf q(Rcurr);
        std::fprintf(f, "%.6f %g %g %g %g %g %g %g\n", (double)timestamp / 1000000.0, currentGlobalCamera(0), currentGlobalCamera(1),
                     currentGlobalCamera(2), q.x, q.y, q.z, q.w);
        std::fclose(f);
    }

    // ---- operator path ------------------------------------------------------------------------------------------
    void allocateBuffers()  // KintinuousTracker.cpp:356-382
    {
        depths_curr_.resize(ICPOdometry::LEVELS);
        vmaps_g_prev_.resize(ICPOdometry::LEVELS);
        nmaps_g_prev_.resize(ICPOdometry::LEVELS);
        vmaps_curr_.resize(ICPOdometry::LEVELS);
        nmaps_curr_.resize(ICPOdometry::LEVELS);
        for (int i = 0; i < ICPOdometry::LEVELS; ++i) {
            const int pyr_rows = Resolution::get().rows() >> i, pyr_cols = Resolution::get().cols() >> i;
            depths_curr_[i].create(pyr_rows, pyr_cols);
            vmaps_g_prev_[i].create(pyr_rows * 3, pyr_cols);
            nmaps_g_prev_[i].create(pyr_rows * 3, pyr_cols);
            vmaps_curr_[i].create(pyr_rows * 3, pyr_cols);
            nmaps_curr_[i].create(pyr_rows * 3, pyr_cols);
        }
        vmap_curr_color.create(Resolution::get().rows(), Resolution::get().cols());
        depthRawScaled_.create(Resolution::get().rows(), Resolution::get().cols());
    }

    void vWrapCopyUpdate()  // KintinuousTracker.cpp:1075-1085
    {
        int* w = &voxelWrap.x;
        int* c = &vWrapCopy.x;
        for (int k = 0; k < 3; ++k) {
            c[k] = w[k];
            if (c[k] < 0) c[k] = N - ((-c[k]) % N);
        }
    }

    void computeGlobalCamera(const kt::Vector3f* tcurr)  // KintinuousTracker.cpp:581-595
    {
        const float vs = Volume::get().getVolumeSize();
        const float voxel = vs / (float)N;
        const int* w = &voxelWrap.x;
        for (int k = 0; k < 3; ++k) {
            currentGlobalCamera(k) = (float)((double)volumeBasis(k) - (double)vs * 0.5);
            currentGlobalCamera(k) += (float)w[k] * voxel;
            if (tcurr) currentGlobalCamera(k) += (*tcurr)(k) - volumeBasis(k);
        }
    }

    void pushDensePose(uint64_t ts, const kt::Matrix3f& R, bool loop)
    {
        kt::Matrix4f pose;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) pose(i, j) = R(i, j);
            pose(i, 3) = currentGlobalCamera(i);
        }
        densePoseGraph.push_back(DensePose(ts, pose, loop));
        latestDensePoseId++;
    }

    void pushSlice(const DeviceArray<PointXYZRGB>& cloud, CloudSlice::Dimension dim, unsigned char* rgb, unsigned short* depth,
                   PlaceRecognitionInput* placeRecognitionFrame = 0)
    {
        CloudSlice::PointCloud* pts = new CloudSlice::PointCloud();
        cloud.download(*pts);
        std::lock_guard<std::mutex> lock(cloudMutex);
        cycledMutex = true;
        sharedCloudSlices.push_back(new CloudSlice(pts, dim, lastOdometry, currentGlobalCamera, rmats_.back(), current_utime,
                                                   dim == CloudSlice::FINAL ? nowMicros() : lagTime, rgb, 0, 0, depth, placeRecognitionFrame));
        cloudSignal.notify_all();
    }

    static int voxelTranslation(float translation, float voxel, int thresh)  // KintinuousTracker.cpp:640-667
    {
        const int f = (int)std::floor(translation / voxel);
        if (f < 0) return (-thresh > f) ? -thresh : f;
        return thresh < f ? thresh : f;
    }

    void processFrameOperators(const DeviceArray2D<unsigned short>& depth_raw, const DeviceArray2D<PixelRGB>& colors, uint64_t timestamp)
    {
        const bool angleColor = !ConfigArgs::get().disableColorAngleWeight;
        const float3 device_volume_size = make_float3(tsdf_volume_->getSize()(0), tsdf_volume_->getSize()(1), tsdf_volume_->getSize()(2));
        if (groundTruth && !groundTruth->preRun(lastRgbImage, lastDepthData, timestamp)) return;   // :460-463
        // pyramid, KintinuousTracker.cpp:465-479: skipped by pure RGB-D odometry without the colour angle weight (nothing reads it)
        if (icp || ConfigArgs::get().useRGBDICP || angleColor) {
            bilateralFilter(depth_raw, depths_curr_[0]);
            for (int i = 1; i < ICPOdometry::LEVELS; ++i) pyrDown(depths_curr_[i - 1], depths_curr_[i]);
            for (int i = 0; i < ICPOdometry::LEVELS; ++i) {
                createVMap(intr(i), depths_curr_[i], vmaps_curr_[i]);
                createNMap(vmaps_curr_[i], nmaps_curr_[i]);
            }
        }

        if (global_time_ == 0) {  // :481-557
            kt::Matrix3f init_Rcam = rmats_.back(), init_Rcam_inv;
            kt::Vector3f init_tcam = tvecs_.back();
            ktSafeCall(kt_host_mat33_inverse(init_Rcam.data(), init_Rcam_inv.data()));
            const int3 emptyVoxel = make_int3(0, 0, 0);
            if (rgbd) rgbd->firstRun(depth_raw, colors);   // :499-502
            integrateTsdfVolume(depth_raw, intr, device_volume_size, kt::dev(init_Rcam_inv), kt::dev(init_tcam),
                                tsdf_volume_->getTsdfTruncDist(), tsdf_volume_->data(), depthRawScaled_, emptyVoxel, color_volume_->data(),
                                colors, nmaps_curr_[0], angleColor);
            for (int i = 0; i < ICPOdometry::LEVELS; ++i)
                tranformMaps(vmaps_curr_[i], nmaps_curr_[i], kt::dev(init_Rcam), kt::dev(init_tcam),
                             vmaps_g_prev_[i], nmaps_g_prev_[i]);
            ++global_time_;
            odom_utime = timestamp;   // :527-528
            pushDensePose(timestamp, init_Rcam, true);
            init_utime.assignValue(timestamp);   // :525-556
            {
                const int n = Resolution::get().numPixels();
                unsigned short* firstDepth = new unsigned short[n];
                std::memcpy(firstDepth, lastDepthData, (size_t)n * 2);
                unsigned char* firstImg = new unsigned char[n * 3];
                std::memcpy(firstImg, lastRgbImage, (size_t)n * 3);
                firstDepthData.assignValue(firstDepth);
                firstRgbImage.assignValue(firstImg);
                if (ConfigArgs::get().vocabFile.size())   // :546-549: the first frame is always sampled
                    addToPlaceRecognition(current_utime, lastPlaceRecognitionTrans, lastPlaceRecognitionRot, frameCompression);
                std::lock_guard<std::mutex> lock(cloudMutex);
                cloudSignal.notify_all();
            }
            return;
        }

        // odometry :564-572
        kt::Matrix3f Rcurr;
        kt::Vector3f tcurr;
        lastOdometry = odometryProvider->getIncrementalTransformation(tcurr, Rcurr, depth_raw, colors, timestamp, lastRgbImage, lastDepthData);
        odom_utime = timestamp;   // :574-575
        rmats_.push_back(Rcurr);
        tvecs_.push_back(tcurr);
        computeGlobalCamera(&tcurr);
        if (ConfigArgs::get().dynamicCube) {  // repositionCube :597-600, 384-442
            const kt::Vector3f vx = tsdf_volume_->getVoxelSize();
            kt_host_reposition_cube(Rcurr.data(), tvecs_.back().data(), Volume::get().getVolumeSize(), vx.data(),
                                    parked ? (ConfigArgs::get().staticMode ? N * 3 : N) : ConfigArgs::get().voxelShift, volumeBasis.data());
        }
        // place-recognition tap :601-624: sample now if the camera has moved enough since the last sample, else with the next slab
        bool shiftSend = false, isLoopPose = false;
        if (ConfigArgs::get().vocabFile.size()) {
            const float place_recognition_movement = 0.15f;   // :76
            if (kt_host_place_recognition_movement(Rcurr.data(), currentGlobalCamera.data(), lastPlaceRecognitionRot.data(), lastPlaceRecognitionTrans.data()) >=
                place_recognition_movement) {
                lastPlaceRecognitionRot = Rcurr;
                lastPlaceRecognitionTrans = currentGlobalCamera;
                addToPlaceRecognition(current_utime, lastPlaceRecognitionTrans, lastPlaceRecognitionRot, frameCompression);
                isLoopPose = true;
            } else {
                shiftSend = true;
            }
        }
        kt::Matrix3f Rcurr_inv;
        ktSafeCall(kt_host_mat33_inverse(Rcurr.data(), Rcurr_inv.data()));

        // shift decision and the three axis blocks :627-833
        const kt::Vector3f voxel = tsdf_volume_->getVoxelSize();
        const int thresh = parked ? INT_MAX : ConfigArgs::get().voxelShift;
        int vt[3];
        for (int k = 0; k < 3; ++k) vt[k] = voxelTranslation(tvecs_.back()(k) - volumeBasis(k), voxel(k), thresh);
        for (int axis = 0; axis < 3; ++axis) {
            vWrapCopyUpdate();
            int lo[3] = {0, 0, 0}, hi[3] = {N, N, N};
            int* w = &voxelWrap.x;
            bool cycled = false;
            CloudSlice::Dimension dim = CloudSlice::XPlus;
            const bool back = vt[axis] <= -thresh;
            if (vt[axis] >= thresh) {
                hi[axis] = vt[axis] + 1 + overlap;
                dim = (CloudSlice::Dimension)(axis * 2);
                cycled = true;
            } else if (back) {
                if (axis == 2) { lo[2] = N + (vt[2] - overlap) - 1; hi[2] = N - 1; }  // the reference's z-minus bounds :805
                else lo[axis] = N + (vt[axis] - overlap);
                dim = (CloudSlice::Dimension)(axis * 2 + 1);
                cycled = true;
            }
            if (!cycled) continue;
            DeviceArray<PointXYZRGB> cloud = tsdf_volume_->fetchCloud(cloud_device_, vWrapCopy, color_volume_->data(), lo[0], hi[0], lo[1],
                                                                      hi[1], lo[2], hi[2], voxelWrap);
            clearAxis(axis, back, w[axis], w[axis] + vt[axis]);
            PlaceRecognitionInput* nextPlaceRecognitionFrame = 0;
            if (shiftSend) {   // :706-717, :762-771, :816-825: the slab leaving the volume takes a sample with it
                lastPlaceRecognitionRot = Rcurr;
                lastPlaceRecognitionTrans = currentGlobalCamera;
                nextPlaceRecognitionFrame = addToPlaceRecognition(current_utime, lastPlaceRecognitionTrans, lastPlaceRecognitionRot, frameCompression);
                isLoopPose = true;
                shiftSend = false;
            }
            // mutexOutCloudBuffer :1156-1208
            pushSlice(cloud, dim, 0, 0, nextPlaceRecognitionFrame);
            const float shift = voxel(axis) * (float)vt[axis];
            tvecs_.back()(axis) -= shift;
            w[axis] += vt[axis];
            tcurr(axis) -= shift;
        }
        vWrapCopyUpdate();

        // integrate + raycast + predicted pyramid :864-899
        integrateTsdfVolume(depth_raw, intr, device_volume_size, kt::dev(Rcurr_inv), kt::dev(tcurr),
                            tsdf_volume_->getTsdfTruncDist(), tsdf_volume_->data(), depthRawScaled_, vWrapCopy, color_volume_->data(), colors,
                            nmaps_curr_[0], angleColor);
        raycast(intr, kt::dev(Rcurr), kt::dev(tcurr), tsdf_volume_->getTsdfTruncDist(), device_volume_size,
                tsdf_volume_->data(), vmaps_g_prev_[0], nmaps_g_prev_[0], vWrapCopy, vmap_curr_color, color_volume_->data());
        if (icp || ConfigArgs::get().useRGBDICP)   // :892-899
            for (int i = 1; i < ICPOdometry::LEVELS; ++i) {
                resizeVMap(vmaps_g_prev_[i - 1], vmaps_g_prev_[i]);
                resizeNMap(nmaps_g_prev_[i - 1], nmaps_g_prev_[i]);
            }
        kt::device::sync();
        ++global_time_;
        lastRotation = Rcurr;
        lastTranslation = tvecs_.back();
        pushDensePose(timestamp, Rcurr, isLoopPose);
    }

    void clearAxis(int axis, bool back, int cur, int next)
    {
        // the reference's call shape (KintinuousTracker.cpp:685-686): data() by value into PtrStep<short> / PtrStep<uchar4>
        switch (axis * 2 + (back ? 1 : 0)) {
            case 0: clearVolumeX(tsdf_volume_->data(), cur, next); clearVolumeXc(color_volume_->data(), cur, next); break;
            case 1: clearVolumeXBack(tsdf_volume_->data(), cur, next); clearVolumeXBackc(color_volume_->data(), cur, next); break;
            case 2: clearVolumeY(tsdf_volume_->data(), cur, next); clearVolumeYc(color_volume_->data(), cur, next); break;
            case 3: clearVolumeYBack(tsdf_volume_->data(), cur, next); clearVolumeYBackc(color_volume_->data(), cur, next); break;
            case 4: clearVolumeZ(tsdf_volume_->data(), cur, next); clearVolumeZc(color_volume_->data(), cur, next); break;
            default: clearVolumeZBack(tsdf_volume_->data(), cur, next); clearVolumeZBackc(color_volume_->data(), cur, next); break;
        }
    }
};
