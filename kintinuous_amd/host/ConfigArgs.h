// ConfigArgs.h -- the command-line options of the reference that reach the tracking + fusion path
// (utils/ConfigArgs.h:36-75, 111-200).  Options of the GUI / backend (-v, -m, -od, -il, ...) are accepted and ignored so
// reference command lines keep working.  Additions: -n <N> volume resolution (the reference's compile-time VOL),
// -w/-h image size, -o <prefix> output prefix (the reference derives saveFile from the log name).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

class ConfigArgs {
  public:
    static const ConfigArgs& get(int argc = 0, char** argv = 0)
    {
        static const ConfigArgs instance(argc, argv);
        return instance;
    }

    static void usage(const std::string argv0)
    {
        std::fprintf(stderr,
                     "Usage: %s [Options]\n"
                     "  -l <log.klg>   log file (raw or zlib depth, raw or JPEG colour)\n"
                     "  -dt <threads>  decode the log ahead on this many threads (compressed logs: ~5 ms of inflate + JPEG per VGA frame; default: hardware threads / 4, at most 8; 0 = inside the read call)\n"
                     "  -c <calib>     calibration file: fx fy cx cy\n"
                     "  -s <metres>    volume size (default 6)\n"
                     "  -t <voxels>    voxel shift threshold (default 14)\n"
                     "  -g <gpu>       device\n"
                     "  -n <N>         volume resolution (default 512)\n"
                     "  -r | -ri       RGB-D odometry | RGB-D + ICP odometry\n"
                     "  -v <vocab>     loop-closure vocabulary: sample frames into placeRecognitionBuffer (the DBoW backend itself is not part of this path)\n"
                     "  -p <file>      ground-truth odometry from a trajectory file (lines utime,x,y,z,qx,qy,qz,qw)\n"
                     "  -fod           fast odometry,  -sm static mode,  -d dynamic cube,  -dc no colour angle weight,  -no no overlap\n"
                     "  -f             flip colours (RGB <-> BGR)\n"
                     "  -tum           write poses with timestamps in seconds (TUM format; the .poses default)\n"
                     "  -o <prefix>    output prefix (default: the log name)\n"
                     "  -cw <weight>   weight cull of the slice processor (default 8),  -nos do not keep overlapping points in the saved cloud\n"
                     "  -pcdraw        debug: the extracted slices as they are into <prefix>.raw.pcd (binary, x y z rgb)\n"
                     "  -pcd           run the CloudSliceProcessor stage behind the tracker and save <prefix>.pcd as the reference does\n"
                     "                 (binary pcl::PointXYZRGBNormal: x y z rgb normal_x normal_y normal_z curvature)\n"
                     "  -ppm           write the final model views: <prefix>_model.ppm, _color.ppm, _depth.pgm\n"
                     "  -rank R -world W -comm <file> [-gk K]   one process per GPU (-g): gather every rank's K most recent dense poses at the end (RCCL; default 1)\n",
                     argv0.c_str());
    }

    std::string calibrationFile, logFile, trajectoryFile, saveFile, vocabFile;
    int gpu, voxelShift, volumeResolution, width, height, totalNumFrames, weightCull, decodeThreads;
    float volumeSize;
    bool staticMode, dynamicCube, flipColors, extractOverlap, saveOverlap, useRGBD, useRGBDICP, disableColorAngleWeight, fastOdometry, help;

  private:
    static bool flag(int argc, char** argv, const char* name)
    {
        for (int i = 1; i < argc; ++i)
            if (std::strcmp(argv[i], name) == 0) return true;
        return false;
    }
    static const char* value(int argc, char** argv, const char* name)
    {
        for (int i = 1; i + 1 < argc; ++i)
            if (std::strcmp(argv[i], name) == 0) return argv[i + 1];
        return 0;
    }

    ConfigArgs(int argc, char** argv)
        : gpu(0), voxelShift(14), volumeResolution(512), width(640), height(480), totalNumFrames(0), weightCull(8), decodeThreads(-1), volumeSize(6.0f)
    {
        const char* v;
        if ((v = value(argc, argv, "-c"))) calibrationFile = v;
        if ((v = value(argc, argv, "-l"))) logFile = v;
        if ((v = value(argc, argv, "-p"))) trajectoryFile = v;
        if ((v = value(argc, argv, "-v"))) vocabFile = v;   // a DBoW vocabulary switches the place-recognition tap on (ConfigArgs.h:121)
        if ((v = value(argc, argv, "-g"))) gpu = std::atoi(v);
        if ((v = value(argc, argv, "-t"))) voxelShift = std::atoi(v);
        if ((v = value(argc, argv, "-n"))) volumeResolution = std::atoi(v);
        if ((v = value(argc, argv, "-w"))) width = std::atoi(v);
        if ((v = value(argc, argv, "-h"))) height = std::atoi(v);
        if ((v = value(argc, argv, "-s"))) volumeSize = (float)std::atof(v);
        if ((v = value(argc, argv, "-fl"))) totalNumFrames = std::atoi(v);
        if ((v = value(argc, argv, "-dt"))) decodeThreads = std::atoi(v);   // shell only: RawLogReader decode-ahead
        if ((v = value(argc, argv, "-cw"))) weightCull = std::atoi(v);   // ConfigArgs.h:118 (default 8)
        staticMode = flag(argc, argv, "-sm");
        dynamicCube = flag(argc, argv, "-d");
        flipColors = flag(argc, argv, "-f");
        extractOverlap = !flag(argc, argv, "-no");
        saveOverlap = !flag(argc, argv, "-nos");   // ConfigArgs.h:153
        useRGBD = flag(argc, argv, "-r");
        useRGBDICP = flag(argc, argv, "-ri");
        disableColorAngleWeight = flag(argc, argv, "-dc");
        fastOdometry = flag(argc, argv, "-fod");
        help = flag(argc, argv, "--help");
        if (useRGBDICP) useRGBD = false;  // ConfigArgs.h: -ri wins over -r
        if ((v = value(argc, argv, "-o"))) saveFile = v;
        else saveFile = logFile;
    }
};
