// klg_tool -- reads a .klg log with the C++ RawLogReader (raw / zlib depth, raw / JPEG colour, -f colour flip, the reference's
// "last frame is never returned" quirk) and prints one line per frame: timestamp, crc32 of the depth bytes, crc32 of the B G R bytes.
// No GPU needed; used by tests/test_jpeg.py.     klg_tool -l log.klg -w W -h H [-f] [-dt threads] [-hold]
// -hold: keep the buffers of the three frames before the current one and check, after every read, that they still hold what they held
// when they were handed out (a frame stays valid for three further grabNext calls, with and without decode-ahead); exit code 4 if not.
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <deque>

#include "RawLogReader.h"

int main(int argc, char** argv)
{
    const ConfigArgs& args = ConfigArgs::get(argc, argv);
    if (args.logFile.empty()) { std::fprintf(stderr, "usage: %s -l log.klg -w W -h H [-f]\n", argv[0]); return 2; }
    Resolution::get(args.width, args.height);
    RawLogReader log(args.logFile);
    const size_t n = (size_t)Resolution::get().numPixels();
    bool ok = true, hold = false;
    for (int i = 1; i < argc; ++i) hold = hold || std::strcmp(argv[i], "-hold") == 0;
    struct Held { const unsigned short* depth; const unsigned char* image; unsigned long cd, ci; };
    std::deque<Held> held;
    int frame = 0;
    while (log.grabNext(ok, frame) && ok) {
        const unsigned long cd = crc32(0L, reinterpret_cast<const Bytef*>(log.decompressedDepth), (uInt)(n * 2));
        const unsigned long ci = crc32(0L, reinterpret_cast<const Bytef*>(log.decompressedImage), (uInt)(n * 3));
        std::printf("%lld %08lx %08lx %d\n", (long long)log.timestamp, cd, ci, log.isCompressed ? 1 : 0);
        if (hold) {
            for (size_t k = 0; k < held.size(); ++k)
                if (crc32(0L, reinterpret_cast<const Bytef*>(held[k].depth), (uInt)(n * 2)) != held[k].cd ||
                    crc32(0L, reinterpret_cast<const Bytef*>(held[k].image), (uInt)(n * 3)) != held[k].ci) {
                    std::fprintf(stderr, "frame buffer handed out %zu reads ago was overwritten\n", held.size() - k);
                    return 4;
                }
            const Held h = {log.decompressedDepth, log.decompressedImage, cd, ci};
            held.push_back(h);
            if (held.size() > 3) held.pop_front();
        }
    }
    return 0;
}
