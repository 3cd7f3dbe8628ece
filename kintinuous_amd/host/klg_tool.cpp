// klg_tool -- reads a .klg log with the C++ RawLogReader (raw / zlib depth, raw / JPEG colour, -f colour flip, the reference's
// "last frame is never returned" quirk) and prints one line per frame: timestamp, crc32 of the depth bytes, crc32 of the B G R bytes.
// No GPU needed; used by tests/test_jpeg.py.     klg_tool -l log.klg -w W -h H [-f]
#include <zlib.h>
#include <cstdio>

#include "RawLogReader.h"

int main(int argc, char** argv)
{
    const ConfigArgs& args = ConfigArgs::get(argc, argv);
    if (args.logFile.empty()) { std::fprintf(stderr, "usage: %s -l log.klg -w W -h H [-f]\n", argv[0]); return 2; }
    Resolution::get(args.width, args.height);
    RawLogReader log(args.logFile);
    const size_t n = (size_t)Resolution::get().numPixels();
    bool ok = true;
    int frame = 0;
    while (log.grabNext(ok, frame) && ok) {
        const unsigned long cd = crc32(0L, reinterpret_cast<const Bytef*>(log.decompressedDepth), (uInt)(n * 2));
        const unsigned long ci = crc32(0L, reinterpret_cast<const Bytef*>(log.decompressedImage), (uInt)(n * 3));
        std::printf("%lld %08lx %08lx %d\n", (long long)log.timestamp, cd, ci, log.isCompressed ? 1 : 0);
    }
    return 0;
}
