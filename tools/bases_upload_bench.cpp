// Host key -> device: time of device_bases(ctx, first, last) over 2^20 normalised G1 / 2^18 G2 points held as host group values
// (what r1cs_gg_ppzksnark_proving_key_hip(ctx, pk, dom) does with each query of a key that came from the reference's generator).
#include <chrono>
#include <cstdio>
#include <nil/crypto3/zk/hip/backend.hpp>
using namespace nil::crypto3::zk::hip;
typedef bls12_381 C;
typedef curve_adapter<C> A;
template <int Group, typename G>
void run(const context &ctx, size_t n, size_t cl) {
    std::vector<A::scalar_value_type> ks(n);
    for (size_t i = 0; i < n; ++i) ks[i] = A::scalar_value_type(3 * i + 7);
    auto dev = device_bases<C, Group>::from_scalars(ctx, ks.begin(), ks.end());
    std::vector<uint64_t> xy(n * 2 * cl);
    std::vector<uint8_t> inf(n);
    check(zkhip_bases_download(ctx.get(), dev.get(), 0, n, xy.data(), inf.data()), "download", ctx.get());
    std::vector<G> pts(n);
    for (size_t i = 0; i < n; ++i) pts[i] = G::from_affine(&xy[i * 2 * cl], inf[i] != 0);
    auto t0 = std::chrono::steady_clock::now();
    device_bases<C, Group> up(ctx, pts.begin(), pts.end());
    ctx.sync();
    auto t1 = std::chrono::steady_clock::now();
    std::vector<uint64_t> back(2 * cl);
    uint8_t binf = 0;
    check(zkhip_bases_download(ctx.get(), up.get(), n - 1, 1, back.data(), &binf), "download", ctx.get());
    bool same = std::equal(back.begin(), back.end(), xy.begin() + (n - 1) * 2 * cl);
    printf("G%d: %zu host points -> device bases (window tables included): %.1f ms, round trip %s\n", Group == ZKHIP_G1 ? 1 : 2, n,
           std::chrono::duration<double, std::milli>(t1 - t0).count(), same ? "ok" : "MISMATCH");
}
int main() {
    context ctx(0);
    run<ZKHIP_G1, A::g1_value_type>(ctx, (size_t)1 << 20, A::g1_coord_limbs);
    run<ZKHIP_G2, A::g2_value_type>(ctx, (size_t)1 << 18, A::g2_coord_limbs);
    return 0;
}
