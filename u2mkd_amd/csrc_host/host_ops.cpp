// Host side of the hottest operators in C++ (autograd nodes included), above the C ABI of libu2mkd_hip.so.
//
// The pipelined KD step is bound by the HOST (NOTES N10: ~62 ms of interpreter + dispatcher time per 64 ms step, every C-ABI call
// short): ~210 forward and ~170 backward applications of three Python autograd Functions -- BatchNorm rows, nn.Linear over point
// features, sparse convolution -- cost 30-100 us each, mostly Python bodies.  The same Functions here: allocation through ATen,
// ONE C-ABI call per pass, torch::autograd::Function nodes; the Python Functions of torchsparse/nn/functional.py remain for
// every configuration this file does not cover (bf16 rows, SyncBatchNorm, non-default arithmetic) and are what the tests compare
// against.  Reference semantics: spnn.BatchNorm / nn.BatchNorm1d (core/models/build_blocks.py:30-31), nn.Linear
// (core/models/semantickitti/spvcnn.py:58-74), spnn.Conv3d (build_blocks.py:21-83).
#include <torch/extension.h>
#include <ATen/hip/HIPContext.h>

#include "../../include/u2mkd_hip.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::Variable;
using torch::autograd::variable_list;
using OptTensor = c10::optional<at::Tensor>;

inline void check(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + " failed (rc=" + std::to_string(rc) + "): " + u2mkd_last_error());
}
inline u2mkd_stream_t cur_stream() { return reinterpret_cast<u2mkd_stream_t>(at::hip::getCurrentHIPStream().stream()); }
inline const float *cf(const OptTensor &t) { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; }
inline float *mf(const OptTensor &t) { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; }
inline at::Tensor undef() { return at::Tensor(); }

// ---------------------------------------------------------------------------------------------------- BatchNorm rows
// functional.BatchNormFunction, fp32 rows: y = [relu](bn(x) [+ res]); statistics / running statistics / step counter inside
// u2mkd_bn_train_forward_res; backward re-derives the ReLU mask from x.
struct BatchNormRows : public torch::autograd::Function<BatchNormRows> {
    static at::Tensor forward(AutogradContext *ctx, at::Tensor x, OptTensor gamma, OptTensor beta, OptTensor running_mean,
                              OptTensor running_var, bool training, double momentum, double eps, bool relu, OptTensor counter,
                              OptTensor res) {
        TORCH_CHECK(x.is_cuda() && x.dim() == 2 && x.scalar_type() == at::kFloat, "batch_norm_rows: x must be a [N, C] fp32 HIP tensor");
        x = x.contiguous();
        const int64_t n = x.size(0);
        const int c = (int)x.size(1);
        at::Tensor r;
        if (res.has_value() && res->defined()) {
            TORCH_CHECK(relu && res->sizes() == x.sizes(), "batch_norm_rows: a residual goes with the ReLU and has x's shape");
            r = res->contiguous();
            if (r.scalar_type() != at::kFloat) r = r.to(at::kFloat);
        }
        auto y = at::empty_like(x);
        auto opts = x.options();
        auto invstd = at::empty({c}, opts);
        at::Tensor mean;
        const float *rp = r.defined() ? r.data_ptr<float>() : nullptr;
        if (training) {
            const int64_t slabs = std::max<int64_t>(u2mkd_bn_num_slabs(n), 1);
            auto partial = at::empty({slabs * 2 * c}, opts);
            mean = at::empty({c}, opts);
            int64_t *cnt = counter.has_value() && counter->defined() ? counter->data_ptr<int64_t>() : nullptr;
            check(u2mkd_bn_train_forward_res(x.data_ptr<float>(), rp, n, c, cf(gamma), cf(beta), (float)eps, (float)momentum,
                                             mf(running_mean), mf(running_var), cnt, relu, partial.data_ptr<float>(),
                                             mean.data_ptr<float>(), invstd.data_ptr<float>(), y.data_ptr<float>(), cur_stream()),
                  "u2mkd_bn_train_forward_res");
        } else {
            TORCH_CHECK(running_mean.has_value() && running_mean->defined(), "batch_norm_rows: eval mode needs running statistics");
            mean = *running_mean;
            check(u2mkd_bn_eval_forward_res(x.data_ptr<float>(), rp, n, c, cf(gamma), cf(beta), (float)eps, cf(running_mean),
                                            cf(running_var), relu, invstd.data_ptr<float>(), y.data_ptr<float>(), cur_stream()),
                  "u2mkd_bn_eval_forward_res");
        }
        ctx->save_for_backward({x, gamma.value_or(undef()), beta.value_or(undef()), mean, invstd, r});
        ctx->saved_data["relu"] = relu;
        ctx->saved_data["training"] = training;
        return y;
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        auto saved = ctx->get_saved_variables();
        const at::Tensor &x = saved[0], &gamma = saved[1], &beta = saved[2], &mean = saved[3], &invstd = saved[4], &res = saved[5];
        at::Tensor dy = grads[0].contiguous();
        if (dy.scalar_type() != at::kFloat) dy = dy.to(at::kFloat);
        const int64_t n = x.size(0);
        const int c = (int)x.size(1);
        auto opts = x.options();
        const int64_t slabs = std::max<int64_t>(u2mkd_bn_num_slabs(n), 1);
        auto partial = at::empty({slabs * 2 * c}, opts);
        auto dgb = at::empty({2, c}, opts);                // (dgamma | dbeta in one allocation)
        auto dx = at::empty_like(x);
        at::Tensor dres = res.defined() ? at::empty_like(x) : at::Tensor();
        check(u2mkd_bn_backward_res(dy.data_ptr<float>(), x.data_ptr<float>(), res.defined() ? res.data_ptr<float>() : nullptr, n, c,
                                    mean.data_ptr<float>(), invstd.data_ptr<float>(), gamma.defined() ? gamma.data_ptr<float>() : nullptr,
                                    beta.defined() ? beta.data_ptr<float>() : nullptr, ctx->saved_data["relu"].toBool(),
                                    ctx->saved_data["training"].toBool(), partial.data_ptr<float>(), dgb[0].data_ptr<float>(),
                                    dgb[1].data_ptr<float>(), dx.data_ptr<float>(), dres.defined() ? dres.data_ptr<float>() : nullptr,
                                    cur_stream()),
              "u2mkd_bn_backward_res");
        return {dx, gamma.defined() ? dgb[0] : at::Tensor(), beta.defined() ? dgb[1] : at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), dres};
    }
};

at::Tensor batch_norm_rows(at::Tensor x, OptTensor gamma, OptTensor beta, OptTensor running_mean, OptTensor running_var, bool training,
                           double momentum, double eps, bool relu, OptTensor counter, OptTensor res) {
    return BatchNormRows::apply(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, counter, res);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "u2mkd_amd host operators (C++ autograd nodes above the C ABI of libu2mkd_hip.so)";
    m.def("batch_norm_rows", &batch_norm_rows, "BatchNorm over the rows of [N, C] (+ ReLU, + residual), fp32",
          py::arg("x"), py::arg("gamma"), py::arg("beta"), py::arg("running_mean"), py::arg("running_var"), py::arg("training"),
          py::arg("momentum"), py::arg("eps"), py::arg("relu"), py::arg("counter"), py::arg("res"));
}
