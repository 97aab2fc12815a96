// mnist_idx.h -- idx-ubyte reader with the reference's selection rule.
//
// Replaces mllib/mnist.h:38-101 (big-endian header, magic 0x803 images / 0x801 labels) and
// mllib::readMNIST (mllib/mnist.h:443-530): keep the first NT images PER LABEL in file order; the
// reference then divides by 255 (mnist.h:495) and fixedL.cc:637-642 by 255 again -- here the raw bytes are
// kept and the feature map is applied on the device (tnml_set_data_u8).
#pragma once
#include <array>
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace tnmlh {

struct Dataset {
    int rows = 0, cols = 0;
    std::vector<uint8_t> pixels;      // [n][rows*cols]
    std::vector<double> gray;         // [n][rows*cols] in byte units, only after reduce() (block means are fractional)
    std::vector<int32_t> labels;      // [n]
    std::vector<long> file_index;     // position in the file (Data::n of mllib/data.h)
    std::array<int, 10> counts{};
    int size() const { return (int)labels.size(); }
    int npix() const { return rows * cols; }
    bool reduced() const { return !gray.empty(); }
    double value(size_t img, size_t j) const { return reduced() ? gray[img * npix() + j] : (double)pixels[img * npix() + j]; }
};

inline uint32_t read_be32(std::ifstream& f) {
    unsigned char b[4];
    f.read(reinterpret_cast<char*>(b), 4);
    if (!f) throw std::runtime_error("idx file truncated in header");
    return (uint32_t(b[0]) << 24) | (uint32_t(b[1]) << 16) | (uint32_t(b[2]) << 8) | uint32_t(b[3]);
}

inline Dataset read_idx(const std::string& image_file, const std::string& label_file, long nt_per_label) {
    std::ifstream fi(image_file, std::ios::binary), fl(label_file, std::ios::binary);
    if (!fi) throw std::runtime_error("Error opening file " + image_file);      // mnist.h:57-60
    if (!fl) throw std::runtime_error("Error opening file " + label_file);
    if (read_be32(fi) != 0x803) throw std::runtime_error("Invalid magic number in " + image_file + " (expected 0x803)");
    const uint32_t ni = read_be32(fi), rows = read_be32(fi), cols = read_be32(fi);
    if (read_be32(fl) != 0x801) throw std::runtime_error("Invalid magic number in " + label_file + " (expected 0x801)");
    const uint32_t nl = read_be32(fl);
    if (ni != nl) throw std::runtime_error("image and label files hold different counts");
    std::vector<uint8_t> lab(nl);
    fl.read(reinterpret_cast<char*>(lab.data()), nl);
    if (!fl) throw std::runtime_error("label file truncated");
    Dataset d; d.rows = (int)rows; d.cols = (int)cols;
    const size_t np = (size_t)rows * cols;
    std::vector<uint8_t> img(np);
    for (uint32_t i = 0; i < ni; ++i) {
        fi.read(reinterpret_cast<char*>(img.data()), np);
        if (!fi) throw std::runtime_error("image file truncated");
        const int l = lab[i];
        if (l < 0 || l > 9) throw std::runtime_error("label outside 0..9");
        if (d.counts[l] >= nt_per_label) continue;                              // mnist.h:487
        d.counts[l] += 1;
        d.pixels.insert(d.pixels.end(), img.begin(), img.end());
        d.labels.push_back(l);
        d.file_index.push_back((long)i);
    }
    return d;
}

// imglen < side: block-mean down-sampling in the manner of image.h:316-346 `reduce` (dead code in the reference, which
// never reads its `imglen` key -- SURVEY.md 8(f)-4): bsize = side/newlen, blocks start at rem = side % bsize, the new
// pixel is the plain mean of the bsize x bsize block, kept as a real number; sites run over the reduced image in the
// file's row-major order.
inline void reduce(Dataset& d, int newlen) {
    if (d.rows != d.cols) throw std::runtime_error("reduce: image is not square");
    if (newlen == d.rows) return;
    if (newlen < 1 || newlen > d.rows) throw std::runtime_error("reduce: imglen must be between 1 and the image side");
    const int side = d.rows, bsize = side / newlen, rem = side % bsize, n = d.size();
    std::vector<double> g((size_t)n * newlen * newlen);
    for (int i = 0; i < n; ++i)
        for (int ny = 0; ny < newlen; ++ny) for (int nx = 0; nx < newlen; ++nx) {
            double avg = 0.; long cnt = 0;
            for (int oy = rem + bsize * ny; oy < rem + bsize * ny + bsize; ++oy)
                for (int ox = rem + bsize * nx; ox < rem + bsize * nx + bsize; ++ox) { avg += d.pixels[((size_t)i * side + oy) * side + ox]; ++cnt; }
            g[((size_t)i * newlen + ny) * newlen + nx] = avg / cnt;
        }
    d.gray = std::move(g); d.pixels.clear(); d.rows = d.cols = newlen;
}

// datadir layout of the reference (mnist.h:244,262,279,297)
inline Dataset read_mnist(const std::string& datadir, bool train, long nt_per_label) {
    const std::string stem = train ? "train" : "t10k";
    return read_idx(datadir + "/" + stem + "-images-idx3-ubyte", datadir + "/" + stem + "-labels-idx1-ubyte", nt_per_label);
}

}  // namespace tnmlh
