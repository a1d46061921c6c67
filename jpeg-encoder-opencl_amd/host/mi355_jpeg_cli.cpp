// mi355-jpeg: PPM in, JFIF out, through the MI355X encode path.
//
//   mi355-jpeg in.ppm out.jpg [-q N] [--mode strict|standard] [--subsample ref420|none|420|444]
//              [--no-cds] [--device K] [--repeat R] [--bits out.bits]
//   mi355-jpeg --batch IN_DIR OUT_DIR [-q N] [--mode ...] [--subsample ...]
//   mi355-jpeg in.ppm out.jpg --stages [--cpu-telemetry FILE]
//
// --stages: additionally run the reference's stage sequence stage by stage (JpegEncoderHost with the
// reference's signature: one GPU kernel per stage function of utils.hpp between an upload and a download) and
// print its nine CPUTelemetry times; with --cpu-telemetry FILE (nine numbers in microseconds, the fields of
// CPUTelemetry in declaration order: CSC CDS levelShift DCT Quant TotalCopy zigZag RLE Huffman -- e.g. the
// reference CPU path's own times) print the reference's "## Speedups: ##" table
// (OpenCLProject_JpegEncoder.cpp:622-629) for the stage-by-stage path and for the fused path.
//
// --batch: every *.ppm of IN_DIR -> OUT_DIR/<name>.jpg.  Runs of files of one size are encoded as one
// batch through mi355_jpeg_pool_encode (frames sharded over all visible GPUs, no collective),
// framed on the host (mi355_jpeg_wrap_jfif).  Same bytes as the one-file form.
//
// --mode strict (default): the reference's arithmetic; --subsample ref420 (default) = its performCDS
// (2x2 chroma means written back at full resolution), none (= --no-cds) skips it.
// --mode standard: a decodable baseline JPEG (not a behaviour of the reference); --subsample 420
// (default there) = real 16x16 MCUs, 444 = one block per component; --restart adds DRI + RSTm markers
// every 64 MCUs.
//
// With no arguments, run from a build/ directory like the reference
// (README.md:43, src/OpenCLProject_JpegEncoder.cpp:320): read ../data/fruit.ppm and
// report the stage times; the reference writes no output file, this tool writes
// ../data/fruit.jpg.  Strict mode reproduces the reference's arithmetic, so the file's
// pixels are not a meaningful picture; its scan bits are the parity artefact.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <string>
#include <vector>

#include "mi355_utils.hpp"

static void usage() {
    std::cout << "usage: mi355-jpeg in.ppm out.jpg [-q 1..100] [--mode strict|standard] [--subsample ref420|none|420|444]\n"
                 "                  [--restart] [--no-cds] [--device K] [--repeat R] [--bits file]\n"
                 "       mi355-jpeg --batch IN_DIR OUT_DIR [-q ..] [--mode ..] [--subsample ..]   (all visible GPUs)\n"
                 "       (no arguments: ../data/fruit.ppm -> ../data/fruit.jpg, like the reference's fixed paths)\n";
}

// --batch: directory in, directory out, through the multi-GPU pool
static int run_batch(const std::string& in_dir, const std::string& out_dir, int quality, unsigned flags) {
    namespace fs = std::filesystem;
    std::vector<std::string> files;
    std::error_code ec;
    for (const auto& e : fs::directory_iterator(in_dir, ec))
        if (e.is_regular_file() && e.path().extension() == ".ppm") files.push_back(e.path().string());
    if (ec || files.empty()) {
        std::cout << "mi355-jpeg: no .ppm files in " << in_dir << std::endl;
        return 1;
    }
    std::sort(files.begin(), files.end());
    fs::create_directories(out_dir, ec);
    mi355_jpeg_pool* pool = nullptr;
    mi355_jpeg_ctx* ctx = nullptr;  // tables for the framing
    int rc = mi355_jpeg_pool_create(nullptr, 0, &pool);
    if (!rc) rc = mi355_jpeg_pool_set_quality(pool, quality);
    if (!rc) rc = mi355_jpeg_create(0, &ctx);
    if (!rc) rc = mi355_jpeg_set_quality(ctx, quality);
    if (rc) {
        std::cout << "mi355-jpeg: " << mi355_jpeg_strerror(rc) << std::endl;
        return 1;
    }
    std::cout << "Batch: " << files.size() << " files, " << mi355_jpeg_pool_workers(pool) << " GPU worker(s)" << std::endl;
    const size_t kMaxBatchBytes = (size_t)2 << 30;  // host memory per batch of frames
    double px_total = 0, enc_seconds = 0;
    const auto t_all = std::chrono::steady_clock::now();
    size_t i = 0;
    int status = 0;
    while (i < files.size() && !status) {
        // a run of consecutive files of one size
        std::vector<uint8_t> frames;
        std::vector<std::string> names;
        size_t W = 0, H = 0;
        while (i < files.size()) {
            ppm_t img;
            if (readPPMImage(files[i].c_str(), &img.width, &img.height, &img.data) == -1) {
                std::cout << "  skipped (not a P6/255 file): " << files[i] << std::endl;
                ++i;
                continue;
            }
            if (names.empty()) W = img.width, H = img.height;
            if (img.width != W || img.height != H || frames.size() + W * H * 3 > kMaxBatchBytes) {
                free(img.data);
                break;  // starts the next run
            }
            frames.insert(frames.end(), (uint8_t*)img.data, (uint8_t*)img.data + W * H * 3);
            names.push_back(files[i]);
            free(img.data);
            ++i;
        }
        if (names.empty()) continue;
        const uint32_t n = (uint32_t)names.size();
        const size_t stride = (mi355_jpeg_scan_bound((uint32_t)W, (uint32_t)H) + 3) & ~(size_t)3;
        std::vector<uint8_t> scans((size_t)n * stride), file;
        std::vector<uint64_t> bits(n);
        double secs = 0;
        rc = mi355_jpeg_pool_encode(pool, frames.data(), (uint32_t)W, (uint32_t)H, n, flags, scans.data(), stride, bits.data(), &secs);
        if (rc) {
            std::cout << "mi355-jpeg: " << mi355_jpeg_strerror(rc) << std::endl;
            status = 1;
            break;
        }
        enc_seconds += secs;
        px_total += (double)n * W * H;
        for (uint32_t f = 0; f < n && !status; ++f) {
            file.resize(2 * (size_t)((bits[f] + 7) / 8) + 4096);
            size_t len = 0;
            rc = mi355_jpeg_wrap_jfif(ctx, scans.data() + (size_t)f * stride, bits[f], (uint32_t)W, (uint32_t)H, flags,
                                      file.data(), file.size(), &len);
            const std::string out = (fs::path(out_dir) / fs::path(names[f]).stem()).string() + ".jpg";
            FILE* fp = rc ? nullptr : fopen(out.c_str(), "wb");
            if (!fp) {
                std::cout << "mi355-jpeg: cannot write " << out << std::endl;
                status = 1;
                break;
            }
            fwrite(file.data(), 1, len, fp);
            fclose(fp);
        }
        std::cout << "  " << n << " x " << W << "x" << H << ": " << secs * 1e3 << " ms encode incl. transfers" << std::endl;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
    if (!status)
        std::cout << "Encoded " << px_total / 1e6 << " Mpixel: " << px_total / 1e6 / (enc_seconds > 0 ? enc_seconds : 1) << " Mpixel/s in the encode calls (PCIe-inclusive), "
                  << px_total / 1e6 / wall << " Mpixel/s wall incl. file I/O" << std::endl;
    mi355_jpeg_destroy(ctx);
    mi355_jpeg_pool_destroy(pool);
    return status;
}

int main(int argc, char** argv) {
    std::string in = "../data/fruit.ppm", out = "../data/fruit.jpg", bits_path;
    int quality = 50, device = 0, repeat = 1, pos = 0;
    bool cds = true, batch = false, restart = false, stages = false;
    std::string cpu_tel_path;
    std::string mode = "strict", subsample;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            usage();
            return 0;
        } else if (a == "-q" && i + 1 < argc) {
            quality = atoi(argv[++i]);
        } else if (a == "--no-cds") {
            cds = false;
        } else if (a == "--batch") {
            batch = true;
        } else if (a == "--restart") {
            restart = true;
        } else if (a == "--stages") {
            stages = true;
        } else if (a == "--cpu-telemetry" && i + 1 < argc) {
            cpu_tel_path = argv[++i];
        } else if (a == "--mode" && i + 1 < argc) {
            mode = argv[++i];
        } else if (a == "--subsample" && i + 1 < argc) {
            subsample = argv[++i];
        } else if (a == "--device" && i + 1 < argc) {
            device = atoi(argv[++i]);
        } else if (a == "--repeat" && i + 1 < argc) {
            repeat = atoi(argv[++i]);
        } else if (a == "--bits" && i + 1 < argc) {
            bits_path = argv[++i];
        } else if (a[0] != '-' && pos == 0) {
            in = a;
            ++pos;
        } else if (a[0] != '-' && pos == 1) {
            out = a;
            ++pos;
        } else {
            usage();
            return 2;
        }
    }
    if (quality < 1 || quality > 100 || repeat < 1) {
        usage();
        return 2;
    }
    unsigned mode_flags = 0;
    if (mode == "strict") {
        if (subsample == "none") cds = false;
        else if (!subsample.empty() && subsample != "ref420") {
            usage();
            return 2;
        }
    } else if (mode == "standard") {
        if (subsample.empty() || subsample == "420") mode_flags = MI355_F_STANDARD | MI355_F_420;
        else if (subsample == "444") mode_flags = MI355_F_STANDARD;
        else {
            usage();
            return 2;
        }
    } else {
        usage();
        return 2;
    }
    if (restart) {  // DRI/RSTm every 64 MCUs: standard mode, one-file form (the markers go in with the device-side stuffing)
        if (!mode_flags || batch) {
            usage();
            return 2;
        }
        mode_flags |= MI355_F_RESTART;
    }
    if (batch) {
        if (pos != 2) {
            usage();
            return 2;
        }
        return run_batch(in, out, quality, mode_flags ? mode_flags : (cds ? MI355_F_CDS : 0u));
    }
    ppm_t img;
    if (readPPMImage(in.c_str(), &img.width, &img.height, &img.data) == -1) return 1;
    std::cout << "Image " << in << ": " << img.width << " x " << img.height << std::endl;
    if (mi355_select(device, quality)) return 1;
    mi355_set_mode(mode_flags);

    std::cout << "\n### MI355X Implementation ###" << std::endl;
    GPUTelemetry tel;
    std::string scan;
    for (int r = 0; r < repeat; ++r)
        if (JpegEncoderDevice(img, &tel, r == repeat - 1 ? &scan : NULL, cds)) return 1;
    std::cout << "Block encode (CSC..RLE/Huffman strings) Time GPU: " << tel.blockEncodeTime << " us\n"
              << "DC heads Time GPU: " << tel.fixupTime << " us\n"
              << "Prefix scan Time GPU: " << tel.scanTime << " us\n"
              << "Bit string emit Time GPU: " << tel.emitTime << " us\n"
              << "Total Time GPU: " << tel.totalTime << " us (wall incl. transfers " << tel.wallTime << " us)\n"
              << "Scan bits: " << scan.size() << std::endl;
    if (!bits_path.empty()) {
        FILE* fp = fopen(bits_path.c_str(), "wb");
        if (fp) {
            fwrite(scan.data(), 1, scan.size(), fp);
            fclose(fp);
        }
    }
    if (writeJpegFile(out.c_str(), img, cds)) return 1;
    std::cout << "Wrote " << out << std::endl;
    if (stages && !mode_flags && cds && quality == 50) {
        // the reference's driver, stage by stage, on a copy (the stage functions work in place)
        ppm_t copy = img;
        copy.data = (rgb_pixel_t*)malloc(img.width * img.height * sizeof(rgb_pixel_t));
        memcpy(copy.data, img.data, img.width * img.height * sizeof(rgb_pixel_t));
        CPUTelemetry g;
        const int rc = JpegEncoderHost(copy, &g);
        free(copy.data);
        if (rc) return 1;
        std::cout << "Stage-by-stage scan " << (mi355_last_scan() == scan ? "equals" : "DIFFERS FROM") << " the fused path's scan ("
                  << mi355_last_scan().size() << " bits)" << std::endl;
        if (mi355_last_scan() != scan) return 1;
        if (!cpu_tel_path.empty()) {
            double c[9];
            FILE* fp = fopen(cpu_tel_path.c_str(), "r");
            int n = 0;
            if (fp) {
                while (n < 9 && fscanf(fp, "%lf", &c[n]) == 1) ++n;
                fclose(fp);
            }
            if (n != 9) {
                std::cout << "mi355-jpeg: --cpu-telemetry needs nine numbers" << std::endl;
                return 1;
            }
            const CPUTelemetry cpu = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]};
            // the reference's table (OpenCLProject_JpegEncoder.cpp:622-629), CPU stage time / GPU stage time
            std::cout << "\n## Speedups: ##" << std::endl;
            std::cout << "Color conversion: " << cpu.CSCTime / g.CSCTime << std::endl;
            std::cout << "Chroma subsampling: " << cpu.CDSTime / g.CDSTime << std::endl;
            std::cout << "Level shifting: " << cpu.levelShiftTime / g.levelShiftTime << std::endl;
            std::cout << "DCT: " << cpu.DCTTime / g.DCTTime << std::endl;
            std::cout << "Quantization: " << cpu.QuantTime / g.QuantTime << std::endl;
            std::cout << "ZigZag: " << cpu.zigZagTime / g.zigZagTime << std::endl;
            std::cout << "RLE: " << cpu.RLETime / g.RLETime << std::endl;
            // beyond the reference's table (its OpenCL path stops at RLE)
            std::cout << "Huffman: " << cpu.HuffmanTime / g.HuffmanTime << std::endl;
            const double cpu_total = c[0] + c[1] + c[2] + c[3] + c[4] + c[5] + c[6] + c[7] + c[8];
            std::cout << "Whole path, fused kernels (device time): " << cpu_total / tel.totalTime << std::endl;
            std::cout << "Whole path, fused kernels (wall incl. transfers): " << cpu_total / tel.wallTime << std::endl;
        }
    } else if (stages) {
        std::cout << "mi355-jpeg: --stages runs the reference's own sequence: strict mode, q 50, chroma averaging on" << std::endl;
        return 2;
    }
    free(img.data);
    return 0;
}
