# CMFHip.jl -- the reference-side binding: a `ccall` layer over libcmf_hip.so (include/cmf_hip.h)
# that plugs the MI355X multiplicative-update rule into CMF.jl's own plugin boundary
# (`abstract type AbstractCFUpdate`, src/algs/alternating.jl:1-8).
#
# NOT EXECUTED IN THIS REPO'S CI: the build container has no Julia.  The tested twin of this file
# is cmf.jl_amd/host.py (same entry points, same argument order, same array layouts).  Julia
# arrays are already in the layout the C ABI expects (column-major Float64), so every call is a
# plain pointer hand-off.
#
# Where to include it: in src/CMF.jl AFTER `include("./algs/pgd.jl")` (src/CMF.jl:35), i.e. after the last algorithm
# file.  The imports below need names that only exist by then: `update_motifs!` / `update_feature_maps!` become
# generic functions in algs/mult.jl (:31), and SquareLoss, MaskedLoss, SquarePenalty, AbsolutePenalty and
# NonnegConstraint are defined in algs/pgd.jl (:35).  (Including it next to algs/mult.jl, as an earlier revision of
# this file said, fails with UndefVarError at the pgd.jl names.)
#
#     results = fit_cnmf(data; L=20, K=32, alg=HIPMultUpdate, max_itr=100)
#
module CMFHip

import ..CMF: AbstractCFUpdate, Tensor
import ..CMF: update_motifs!, update_feature_maps!                                   # generic functions: algs/mult.jl
import ..CMF: SquareLoss, AbsoluteLoss, MaskedLoss, SquarePenalty, AbsolutePenalty,    # types: algs/pgd.jl
              NonnegConstraint, UnitNormConstraint

const LIBCMF = get(ENV, "LIBCMF_HIP", "libcmf_hip.so")

# Regulariser keywords: HEAD's rules read `l1W, l2W, l1H, l2H` (src/algs/mult.jl:23,42; hals.jl:31,37); README.md:44-52
# and BASELINE's `fit_cnmf(data; ..., l1_H=0.1, l2_W=0.5)` spell them `l1_W, l2_W, l1_H, l2_H`, which HEAD's rules let
# fall into `kwargs...` and silently ignore.  The GPU rules accept both; giving both spellings of one weight is an error.
function reg(kwargs, head::Symbol, readme::Symbol, head_value)
    haskey(kwargs, readme) || return Float64(head_value)
    head_value == 0 || error("regulariser given twice: $head and $readme")
    return Float64(kwargs[readme])
end

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:cmf_last_error, LIBCMF), Cstring, ()))
    error("libcmf_hip error $rc: $msg")
end

"""
    HIPMultUpdate(data, W, H; device=0)
    HIPMultUpdate(data, W, H; devices=0:7)

Drop-in for `MultUpdate(data, W, H)` (src/algs/mult.jl:11-20).  Uploads `data`, `W`, `H`; the
rule's scratch (est, numW, denomW, numH, denomH) lives on the GPU.  W and H stay device-resident
between calls and are written back into the caller's arrays after every `update_feature_maps!`
(the reference mutates W and H in place; mult.jl:37-38, :51-52).

With `devices` the rule is a T-sharded group on several GPUs of the node (`cmf_create_multi`): this one Julia
task keeps making the same two calls per iteration (alternating.jl:52,54) and the library runs the sharded
iteration -- ONE RCCL all-reduce of [numW | denomW | loss tail | H halos] per iteration (round 6; the Gram form and PGD keep an
H-halo all-gather of their own), enqueued by one worker thread
per GPU inside the library.  `transport` (include/cmf_hip.h: CMF_COMM_*): 0 = RCCL for distinct devices (default),
4 = direct peer access over xGMI instead of RCCL (opt-in).  `set_option!(rule, "allreduce_overlap", 1)` etc. forward to
cmf_set_option.
"""
mutable struct HIPMultUpdate <: AbstractCFUpdate
    handle::Ptr{Cvoid}
    data_norm::Float64
    sync_every_call::Bool
    # The reference's rules READ the W and H they are handed (mult.jl:23,42).  Under sync_every_call every rule call fingerprints
    # its arguments (cmf_fingerprint) and compares with what this rule last read from / wrote into the caller's arrays; other
    # arrays, or arrays the caller has edited, are uploaded first (`reuploads` counts them).
    verify_args::Symbol      # :sample (one 64-byte line per 4 KB: bulk edits are seen, a single poked element may not be), :full, :none
    strict_inplace::Bool     # update_motifs! also writes W back (synchronous download), so edits between the two calls are honoured
    seen_W::UInt64           # fingerprints of the caller's arrays as this rule last read or wrote them (contents only: `fit`
    seen_H::UInt64           # deep-copies the initial factors, alternating.jl:33-34 -- equal arrays elsewhere are the same factors)
    w_pending::Bool          # update_motifs! has run and W has not been written back yet
    reuploads::Int
end

function fingerprint(rule::HIPMultUpdate, a::Array{Float64})
    fp = Ref{UInt64}(0)
    check(ccall((:cmf_fingerprint, LIBCMF), Cint, (Ptr{Float64}, Int64, Int64, Ref{UInt64}),
                a, length(a), rule.verify_args == :full ? 1 : 64, fp))
    return fp[]
end

function HIPMultUpdate(data::Matrix{Float64}, W::Tensor{Float64}, H::Matrix{Float64};
                       device::Integer=parse(Int, get(ENV, "LOCAL_RANK", "0")), devices=nothing,
                       transport::Integer=0, sync_every_call::Bool=true, verify_args::Symbol=:sample,
                       strict_inplace::Bool=false)
    K, N, L = size(W)
    T = size(data, 2)
    size(data, 1) == N || throw(DimensionMismatch("data has $(size(data,1)) rows, W has N=$N"))
    size(H) == (K, T) || throw(DimensionMismatch("H must be $K x $T"))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if devices === nothing
        check(ccall((:cmf_create, LIBCMF), Cint,
                    (Ref{Ptr{Cvoid}}, Cint, Int64, Int64, Int64, Int64, Ptr{Float64}),
                    h, device, N, T, K, L, data))
    else
        devs = Cint[d for d in devices]
        check(ccall((:cmf_create_multi, LIBCMF), Cint,
                    (Ref{Ptr{Cvoid}}, Cint, Ptr{Cint}, Cint, Int64, Int64, Int64, Int64, Ptr{Float64}),
                    h, length(devs), devs, transport, N, T, K, L, data))   # 0 = CMF_COMM_AUTO: RCCL for distinct devices
    end
    check(ccall((:cmf_set_factors, LIBCMF), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[], W, H))
    ss = Ref{Float64}(0.0)
    check(ccall((:cmf_get_data_sumsq, LIBCMF), Cint, (Ptr{Cvoid}, Ref{Float64}), h[], ss))
    rule = HIPMultUpdate(h[], sqrt(ss[]), sync_every_call, verify_args, strict_inplace, UInt64(0), UInt64(0), false, 0)
    rule.seen_W = fingerprint(rule, W)   # what cmf_set_factors has just read
    rule.seen_H = fingerprint(rule, H)
    finalizer(r -> (r.handle != C_NULL && ccall((:cmf_destroy, LIBCMF), Cint, (Ptr{Cvoid},), r.handle); r.handle = C_NULL), rule)
    return rule
end

# The reference's rules mutate W and H in place (mult.jl:37-38, :51-52) and `fit` hands the same arrays to every call
# (alternating.jl:51-54).  With `sync_every_call` (default) the update_feature_maps! that follows writes the new factors into
# W and H before it returns: cmf_arm_writeback borrows the two arrays until that call returns (hence GC.@preserve around both),
# and the download runs underneath the call's own kernels -- W during the H phase's first contraction, H during the loss
# conv, Float64 widening on helper threads -- instead of a 23 MB synchronous cmf_get_factors per iteration
# (INTEGRATION.md section 2 quotes the measured cost of both).
function arm_writeback(rule::HIPMultUpdate, W::Array{Float64}, H::Array{Float64})
    rule.sync_every_call || return
    check(ccall((:cmf_arm_writeback, LIBCMF), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), rule.handle, W, H))
end

# The rule reads its arguments (mult.jl:23,42): arrays this rule has not seen in this state are uploaded first.
# W reaches the caller's array when update_feature_maps! returns (the write-back), not when update_motifs! does -- in between
# the caller's W is the one update_motifs! started from, so an edit made there is an edit of an outdated W: a clear error,
# unless the rule was built with strict_inplace=true (update_motifs! then downloads W synchronously, + ~0.5 ms at config 2).
# INTEGRATION.md section 3 states the contract; tests/test_dropin_contract.py pins it on the Python twin.
function sync_args!(rule::HIPMultUpdate, W::Array{Float64}, H::Array{Float64})
    (rule.sync_every_call && rule.verify_args != :none) || return
    nowW, nowH = fingerprint(rule, W), fingerprint(rule, H)
    dW, dH = nowW != rule.seen_W, nowH != rule.seen_H
    if dW && rule.w_pending
        error("W was modified (or another array was passed) between update_motifs! and update_feature_maps!: with " *
              "sync_every_call the caller's W holds the motifs update_motifs! started from until update_feature_maps! " *
              "returns.  Build the rule with strict_inplace=true, or call upload!(rule, W, H).")
    end
    if dW || dH
        check(ccall((:cmf_set_factors, LIBCMF), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), rule.handle,
                    dW ? pointer(W) : Ptr{Float64}(C_NULL), dH ? pointer(H) : Ptr{Float64}(C_NULL)))
        rule.reuploads += 1
    end
    rule.seen_W, rule.seen_H = nowW, nowH
end

function after_motifs!(rule::HIPMultUpdate, W::Array{Float64})
    (rule.sync_every_call && rule.verify_args != :none) || return
    if rule.strict_inplace
        check(ccall((:cmf_get_factors, LIBCMF), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), rule.handle, W, Ptr{Float64}(C_NULL)))
        rule.seen_W = fingerprint(rule, W)
    else
        rule.w_pending = true
    end
end

function after_feature_maps!(rule::HIPMultUpdate, W::Array{Float64}, H::Array{Float64})
    (rule.sync_every_call && rule.verify_args != :none) || return
    rule.seen_W, rule.seen_H = fingerprint(rule, W), fingerprint(rule, H)   # (the write-back has just filled them)
    rule.w_pending = false
end

"upload!(rule, W, H): make W, H the resident factors (cmf_set_factors); the next rule call takes its arguments as they are."
function upload!(rule::HIPMultUpdate, W::Array{Float64}, H::Array{Float64})
    check(ccall((:cmf_set_factors, LIBCMF), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), rule.handle, W, H))
    rule.seen_W, rule.seen_H = fingerprint(rule, W), fingerprint(rule, H)
    rule.w_pending = false
end

# update_motifs!(rule, data, W, H; l1W=0, l2W=0)  -- src/algs/mult.jl:23-39, called at alternating.jl:52
function update_motifs!(rule::HIPMultUpdate, data, W, H; l1W=0, l2W=0, kwargs...)
    GC.@preserve W H begin
        sync_args!(rule, W, H)
        check(ccall((:cmf_update_motifs, LIBCMF), Cint, (Ptr{Cvoid}, Float64, Float64), rule.handle,
                    reg(kwargs, :l1W, :l1_W, l1W), reg(kwargs, :l2W, :l2_W, l2W)))
        after_motifs!(rule, W)
    end
    return W
end

# update_feature_maps!(rule, data, W, H; l1H=0, l2H=0) -> loss  -- src/algs/mult.jl:42-58, alternating.jl:54
function update_feature_maps!(rule::HIPMultUpdate, data, W, H; l1H=0, l2H=0, kwargs...)
    loss = Ref{Float64}(0.0)
    GC.@preserve W H begin
        sync_args!(rule, W, H)
        arm_writeback(rule, W, H)
        check(ccall((:cmf_update_feature_maps, LIBCMF), Cint, (Ptr{Cvoid}, Float64, Float64, Ref{Float64}),
                    rule.handle, reg(kwargs, :l1H, :l1_H, l1H), reg(kwargs, :l2H, :l2_H, l2H), loss))
        after_feature_maps!(rule, W, H)
    end
    return loss[]
end

"""
    HIPHALSUpdate(data, W, H; device=0)

Drop-in for `HALSUpdate(data, W, H)` (src/algs/hals.jl:18-28) on the same handle type: the K*L column
updates of W and the K*T entry updates of H run on the GPU in the reference's Gauss-Seidel order.
"""
mutable struct HIPHALSUpdate <: AbstractCFUpdate
    inner::HIPMultUpdate
end
HIPHALSUpdate(data, W, H; kwargs...) = HIPHALSUpdate(HIPMultUpdate(data, W, H; kwargs...))

# update_motifs!(rule::HALSUpdate, ...; l1W=0, l2W=0)  -- src/algs/hals.jl:31-34
function update_motifs!(rule::HIPHALSUpdate, data, W, H; l1W=0, l2W=0, kwargs...)
    GC.@preserve W H begin
        sync_args!(rule.inner, W, H)
        check(ccall((:cmf_hals_update_motifs, LIBCMF), Cint, (Ptr{Cvoid}, Float64, Float64), rule.inner.handle,
                    reg(kwargs, :l1W, :l1_W, l1W), reg(kwargs, :l2W, :l2_W, l2W)))
        after_motifs!(rule.inner, W)
    end
    return W
end

# update_feature_maps!(rule::HALSUpdate, ...; l1H=0, l2H=0) -> loss  -- src/algs/hals.jl:37-42
function update_feature_maps!(rule::HIPHALSUpdate, data, W, H; l1H=0, l2H=0, kwargs...)
    loss = Ref{Float64}(0.0)
    GC.@preserve W H begin
        sync_args!(rule.inner, W, H)
        arm_writeback(rule.inner, W, H)
        check(ccall((:cmf_hals_update_feature_maps, LIBCMF), Cint, (Ptr{Cvoid}, Float64, Float64, Ref{Float64}),
                    rule.inner.handle, reg(kwargs, :l1H, :l1_H, l1H), reg(kwargs, :l2H, :l2_H, l2H), loss))
        after_feature_maps!(rule.inner, W, H)
    end
    return loss[]
end

"""
    HIPPGDUpdate(data, W, H; device=0)

Drop-in for `PGDUpdate(data, W, H)` (src/algs/pgd.jl:112-155) with `SquareLoss()`, `AbsoluteLoss()` or
`MaskedLoss(either, mask)`, `SquarePenalty` / `AbsolutePenalty` lists and `NonnegConstraint()` /
`UnitNormConstraint()` / `nothing`.
The step-size state (stepW, stepH, cur_loss) lives in the library handle.
"""
mutable struct HIPPGDUpdate <: AbstractCFUpdate
    inner::HIPMultUpdate
    mask_id::UInt            # objectid of the mask currently resident on the device (0 = none)
    loss_kind::Cint          # 0 SquareLoss, 1 AbsoluteLoss (cmf_pgd_set_loss)
end
# `devices=0:7` shards T over several GPUs like HIPMultUpdate does (one all-reduce of the partial gradW per iteration;
# MaskedLoss masks are cut along T by the library)
function HIPPGDUpdate(data, W, H; kwargs...)
    rule = HIPPGDUpdate(HIPMultUpdate(data, W, H; kwargs...), UInt(0), Cint(0))
    check(ccall((:cmf_pgd_reset, LIBCMF), Cint, (Ptr{Cvoid},), rule.inner.handle))
    return rule
end

penalty_weights(pens) = (sum(Float64[p.weight for p in pens if p isa SquarePenalty]),
                         sum(Float64[p.weight for p in pens if p isa AbsolutePenalty]))
# README-style regularisers (`l1_W`, `l2_W`, `l1_H`, `l2_H`) do NOT reach the PGD rule: the reference's PGD methods take
# penalties only through `penaltiesW` / `penaltiesH` and swallow every other keyword in `kwargs...` (pgd.jl:158-202), and
# so do these (and PGDUpdate in cmf.jl_amd/host.py): `fit_cnmf(data; alg=:pgd, l2_W=0.5)` gives the factors of
# `fit_cnmf(data; alg=:pgd)` here, in the Python mirror and in the reference alike.
# 0 none, 1 NonnegConstraint (pgd.jl:92-96), 2 UnitNormConstraint (pgd.jl:100-110)
nonneg_flag(c) = c === nothing ? Cint(0) : (c isa NonnegConstraint ? Cint(1) :
                 (c isa UnitNormConstraint ? Cint(2) : error("unsupported constraint")))

function select_loss!(rule::HIPPGDUpdate, loss_func)
    base = loss_func isa MaskedLoss ? loss_func.loss : loss_func             # pgd.jl:58-70
    kind = base isa SquareLoss ? Cint(0) : (base isa AbsoluteLoss ? Cint(1) :
           error("HIPPGDUpdate supports SquareLoss(), AbsoluteLoss() and MaskedLoss of either"))
    if kind != rule.loss_kind
        check(ccall((:cmf_pgd_set_loss, LIBCMF), Cint, (Ptr{Cvoid}, Cint), rule.inner.handle, kind))
        rule.loss_kind = kind
    end
    if loss_func isa MaskedLoss
        id = objectid(loss_func.mask)
        if id != rule.mask_id
            m = Matrix{Float64}(loss_func.mask)
            check(ccall((:cmf_set_mask, LIBCMF), Cint, (Ptr{Cvoid}, Ptr{Float64}), rule.inner.handle, m))
            rule.mask_id = id
        end
    else
        rule.mask_id == 0 || check(ccall((:cmf_set_mask, LIBCMF), Cint, (Ptr{Cvoid}, Ptr{Float64}), rule.inner.handle, C_NULL))
        rule.mask_id = UInt(0)
    end
end

# update_motifs!(rule::PGDUpdate, ...; loss_func, constrW, penaltiesW)  -- src/algs/pgd.jl:158-177
function update_motifs!(rule::HIPPGDUpdate, data, W, H; loss_func=SquareLoss(), constrW=NonnegConstraint(),
                        penaltiesW=[SquarePenalty(1)], kwargs...)
    select_loss!(rule, loss_func)
    sq, ab = penalty_weights(penaltiesW)
    GC.@preserve W H begin
        sync_args!(rule.inner, W, H)
        check(ccall((:cmf_pgd_update_motifs, LIBCMF), Cint, (Ptr{Cvoid}, Float64, Float64, Cint),
                    rule.inner.handle, sq, ab, nonneg_flag(constrW)))
        after_motifs!(rule.inner, W)
    end
    return W
end

# update_feature_maps!(rule::PGDUpdate, ...; loss_func, constrH, penaltiesH) -> loss  -- src/algs/pgd.jl:180-202
function update_feature_maps!(rule::HIPPGDUpdate, data, W, H; loss_func=SquareLoss(), constrH=NonnegConstraint(),
                              penaltiesH=[], kwargs...)
    select_loss!(rule, loss_func)
    sq, ab = penalty_weights(penaltiesH)
    loss = Ref{Float64}(0.0)
    GC.@preserve W H begin
        sync_args!(rule.inner, W, H)
        arm_writeback(rule.inner, W, H)
        check(ccall((:cmf_pgd_update_feature_maps, LIBCMF), Cint, (Ptr{Cvoid}, Float64, Float64, Cint, Ref{Float64}),
                    rule.inner.handle, sq, ab, nonneg_flag(constrH), loss))
        after_feature_maps!(rule.inner, W, H)
    end
    return loss[]
end

"""
    ALGORITHMS

README.md:16,30-33 selects the algorithm with a symbol (`alg=:mult`, `:hals`); HEAD's table for that is commented out
(src/model.jl:3-8) and `fit_cnmf` calls `alg(data, W_init, H_init)` (model.jl:78-82), so a symbol fails there with a
MethodError.  INTEGRATION.md section 2b shows the five-line patch to model.jl that looks `alg` up here first.
"""
const ALGORITHMS = Dict{Symbol,Any}(
    :mult => HIPMultUpdate,
    :hals => HIPHALSUpdate,
    :pgd => HIPPGDUpdate,
)

"""
    iterate!(rule, n; l1W=0, l2W=0, l1H=0, l2H=0, eval_mode=false) -> losses

`n` x (`update_motifs!`; `update_feature_maps!`) back to back (alternating.jl:51-54) in one ccall (`cmf_iterate`): the
n losses; the host never stalls the device between iterations.  W and H are not copied back (use `download!`).
"""
function iterate!(rule::HIPMultUpdate, n::Integer; l1W=0, l2W=0, l1H=0, l2H=0, eval_mode::Bool=false)
    losses = zeros(n)
    check(ccall((:cmf_iterate, LIBCMF), Cint,
                (Ptr{Cvoid}, Int64, Cint, Float64, Float64, Float64, Float64, Ptr{Float64}, Ptr{Float64}),
                rule.handle, n, eval_mode ? 1 : 0, l1W, l2W, l1H, l2H, losses, C_NULL))
    return losses
end

"Library option (include/cmf_hip.h, cmf_set_option): \"reuse_est\", \"speculate\", \"gram\", \"small_k\", \"allreduce_overlap\", \"enqueue_threads\", ..."
set_option!(rule::HIPMultUpdate, name::AbstractString, value::Integer) =
    check(ccall((:cmf_set_option, LIBCMF), Cint, (Ptr{Cvoid}, Cstring, Cint), rule.handle, name, value))

"Library identification: \"cmf_hip gfx950 <version> abi=<n> src=<digest of the sources it was built from>\"."
version() = unsafe_string(ccall((:cmf_version, LIBCMF), Cstring, ()))

"Block until everything the rule has enqueued has finished (all devices of a `devices=` group)."
synchronize(rule::HIPMultUpdate) = check(ccall((:cmf_synchronize, LIBCMF), Cint, (Ptr{Cvoid},), rule.handle))

"Write the device-resident factors into W and H (needed only with `sync_every_call=false`)."
function download!(rule::HIPMultUpdate, W::Tensor{Float64}, H::Matrix{Float64})
    check(ccall((:cmf_get_factors, LIBCMF), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), rule.handle, W, H))
    return W, H
end

# Stand-alone primitives: src/common.jl:17-34, :62-81
function tensor_conv(W::Tensor{Float64}, H::Matrix{Float64}; device::Integer=0)
    K, N, L = size(W); T = size(H, 2)
    est = zeros(N, T)
    check(ccall((:cmf_tensor_conv, LIBCMF), Cint,
                (Cint, Int64, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), device, N, T, K, L, W, H, est))
    return est
end

function tensor_transconv(W::Tensor{Float64}, X::Matrix{Float64}; device::Integer=0)
    K, N, L = size(W); T = size(X, 2)
    out = zeros(K, T)
    check(ccall((:cmf_tensor_transconv, LIBCMF), Cint,
                (Cint, Int64, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), device, N, T, K, L, W, X, out))
    return out
end

# init_rand(data, L, K): src/model.jl:113-125 (portable RNG; `seed` as in fit_cnmf's kwarg)
function init_rand(data::Matrix{Float64}, L::Integer, K::Integer; seed::Integer=rand(UInt64), device::Integer=0)
    N, T = size(data)
    W = zeros(K, N, L); H = zeros(K, T)
    check(ccall((:cmf_init_rand, LIBCMF), Cint,
                (Cint, Int64, Int64, Int64, Int64, UInt64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                device, N, T, K, L, seed % UInt64, data, W, H))
    return W, H
end

# gen_synthetic(N=, T=): README.md:14 / datasets/synthetic.jl:29-61
function gen_synthetic(; N=100, T=500, K=3, L=20, alpha=0.1, p_h=0.5, sigma=0.2, noise_scale=1.0, seed=1234, device::Integer=0)
    data = zeros(N, T)
    check(ccall((:cmf_gen_synthetic, LIBCMF), Cint,
                (Cint, Int64, Int64, Int64, Int64, Float64, Float64, Float64, Float64, UInt64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                device, N, T, K, L, alpha, p_h, sigma, noise_scale, seed % UInt64, data, C_NULL, C_NULL))
    return data
end

end # module
