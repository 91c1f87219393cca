# EnlsipHIP.jl — thin ccall glue that plugs libenlsip_gn.so underneath Enlsip.jl's
# Gauss-Newton subproblem (gn_search_direction + the QR lines of update_working_set), leaving
# CnlsModel / solve! and everything above the seam unchanged.
#
# STATUS: written against include/enlsip_gn.h and the reference sources, but NOT executed:
# no Julia toolchain exists in the build container or on the GPU box (SURVEY.md §0, §8c).  The
# same C ABI is exercised end to end by the ctypes host mirror (enlsip.jl_amd/python) and
# tests/test_gpu_parity.py; INTEGRATION.md shows how a maintainer wires this file in.
#
# Seam (reference lines):
#   update_working_set   src/enlsip_functions.jl:686-795   (QR lines :700, 722-725, 740-743, 758-762, 768-771, 786-789)
#   gn_search_direction  src/enlsip_functions.jl:206-234
#   sub_search_direction src/enlsip_functions.jl:116-153   (re-entered at :1253)
module EnlsipHIP

using LinearAlgebra

const LIB = get(ENV, "ENLSIP_GN_LIB", joinpath(@__DIR__, "..", "lib", "libenlsip_gn.so"))

const FACTOR_A = Cint(0)
const FACTOR_L11 = Cint(1)
const FACTOR_J2 = Cint(2)

struct Opts
    device::Int32
    flags::Int32
    panel_width::Int32
    tile_rows::Int32
    stream::Ptr{Cvoid}
end

struct Info
    rankA::Int64
    rankJ2::Int64
    code::Int64
    dimA::Int64
    dimJ2::Int64
    status::Int64
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(; device::Integer=-1, flags::Integer=0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        opts = Ref(Opts(Int32(device), Int32(flags), Int32(0), Int32(0), C_NULL))
        rc = ccall((:enlsip_gn_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{Opts}), ref, opts)
        rc == 0 || error("enlsip_gn_create failed with code $rc: " *
                         unsafe_string(ccall((:enlsip_gn_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
        h = new(ref[])
        finalizer(x -> ccall((:enlsip_gn_destroy, LIB), Cint, (Ptr{Cvoid},), x.ptr), h)
        return h
    end
end

function check(h::Handle, rc::Integer)
    rc == 0 && return
    msg = unsafe_string(ccall((:enlsip_gn_last_error, LIB), Cstring, (Ptr{Cvoid},), h.ptr))
    error("libenlsip_gn error $rc: $msg")
end

# QRPivoted-like shim backed by the device-resident factors of the last solve on `h`
# (valid until the next solve).  Supports exactly what the reference's consumers use:
# F.R, F.p, F.P, F.Q' * v, F.Q * v  (src/enlsip_functions.jl:461-537, 1118-1291).
struct DeviceQR
    h::Handle
    which::Cint
    rows::Int      # length of vectors Q acts on
end

function factor_shape(F::DeviceQR)
    r = Ref{Int64}(0); c = Ref{Int64}(0)
    check(F.h, ccall((:enlsip_gn_factor_shape, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Ref{Int64}, Ref{Int64}),
                     F.h.ptr, F.which, 0, r, c))
    return r[], c[]
end

function Base.getproperty(F::DeviceQR, s::Symbol)
    if s === :R
        r, c = factor_shape(F)
        R = zeros(Float64, max(r, 1), c)
        GC.@preserve R check(F.h, ccall((:enlsip_gn_get_R, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Int64),
                                        getfield(F, :h).ptr, getfield(F, :which), 0, R, max(r, 1)))
        return R[1:r, :]
    elseif s === :p
        _, c = factor_shape(F)
        p = zeros(Int64, c)
        c > 0 && GC.@preserve p check(getfield(F, :h), ccall((:enlsip_gn_get_jpvt, LIB), Cint,
                                      (Ptr{Cvoid}, Cint, Int64, Ptr{Int64}), getfield(F, :h).ptr, getfield(F, :which), 0, p))
        return p
    elseif s === :P
        p = F.p
        n = length(p)
        P = zeros(Float64, n, n)
        for i in 1:n
            P[p[i], i] = 1.0
        end
        return P
    elseif s === :Q
        return DeviceQ(F, false)
    else
        return getfield(F, s)
    end
end

struct DeviceQ
    F::DeviceQR
    adj::Bool
end
Base.adjoint(Q::DeviceQ) = DeviceQ(Q.F, !Q.adj)

function Base.:*(Q::DeviceQ, v::AbstractVector{Float64})
    out = Vector{Float64}(v)
    h = getfield(Q.F, :h)
    GC.@preserve out begin
        rc = Q.adj ?
            ccall((:enlsip_gn_apply_qt, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}), h.ptr, getfield(Q.F, :which), 0, out) :
            ccall((:enlsip_gn_apply_q, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}), h.ptr, getfield(Q.F, :which), 0, out)
        check(h, rc)
    end
    return out
end

# M * F.Q and M * F.Q' — what the reference's consumers write as `J * F_A.Q` (src/enlsip_functions.jl:384, :526, :1249), so that
# those lines run UNMODIFIED on a DeviceQR.  `F_A.Q` applied from the right to a matrix with as many rows as the J of the last
# solve is one launch of the library's J*Q1 kernel on the resident reflectors (enlsip_gn_matrix_times_QA: M is staged, nothing
# about it needs to be resident); every other combination (F_L11 / F_J2, adjoints, other row counts) falls back to one
# `Q' * row` per row of M — correct for any M, slow for tall ones, and not on any path the reference takes.
function Base.:*(M::AbstractMatrix{Float64}, Q::DeviceQ)
    F = Q.F
    h = getfield(F, :h)
    rows, cols = size(M)
    cols == getfield(F, :rows) || throw(DimensionMismatch("matrix has $cols columns, Q acts on vectors of length $(getfield(F, :rows))"))
    if getfield(F, :which) == FACTOR_A && !Q.adj
        Md = Matrix{Float64}(M)
        out = zeros(Float64, rows, cols)
        rc = GC.@preserve Md out ccall((:enlsip_gn_matrix_times_QA, LIB), Cint,
                                       (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64),
                                       h.ptr, 0, rows, Md, max(rows, 1), out, max(rows, 1))
        rc == 0 && return out
        rc == -3 || check(h, rc)         # -3: rows differ from the resident plan's m — generic path below
    end
    # (M Q)' = Q' M'  and  (M Q')' = Q M': one vector application per row of M
    out = Matrix{Float64}(undef, rows, cols)
    Qt = DeviceQ(F, !Q.adj)
    for i in 1:rows
        out[i, :] = Qt * Vector{Float64}(M[i, :])
    end
    return out
end

# J * F_A.Q  (src/enlsip_functions.jl:219, :526, :1249): served from the device instead of a second dormqr
function jq1(h::Handle, m::Integer, n::Integer)
    out = zeros(Float64, m, n)
    GC.@preserve out check(h, ccall((:enlsip_gn_get_JQ1, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64), h.ptr, 0, out, m))
    return out
end

"""
    gn_search_direction_hip!(h, J, rx, A_active, cx, ε_rank, current_iter) -> p_gn, F_A, F_L11, F_J2

Replaces, in `update_working_set`, every occurrence of

    F_A = qr(C.A', ColumnNorm()); rankA = pseudo_rank(diag(F_A.R), ε_rank)
    F_L11 = qr(F_A.R', ColumnNorm())
    p_gn[:], F_J2 = gn_search_direction(J, rx, C.cx, F_A, F_L11, rankA, W.t, ε_rank, iter_k)

(src/enlsip_functions.jl:722-725, 740-743, 758-762, 768-771, 786-789) by one call into the HIP
library, and writes the same `Iteration` fields gn_search_direction writes (:226-231).
`A_active` is `C.A` (t x n); its transpose is materialised because the ABI wants `C.A'` column-major.
"""
function gn_search_direction_hip!(h::Handle, J::Matrix{Float64}, rx::Vector{Float64}, A_active::Matrix{Float64},
                                  cx::Vector{Float64}, ε_rank::Float64, current_iter)
    m, n = size(J)
    t = size(A_active, 1)
    At = Matrix{Float64}(transpose(A_active))          # n x t, column-major
    p = zeros(Float64, n); b = zeros(Float64, t); d = zeros(Float64, m)
    info = Ref(Info(0, 0, 0, 0, 0, 0))
    jA = zeros(Int64, t); jL = zeros(Int64, min(n, t)); jJ = zeros(Int64, n)
    GC.@preserve J rx At cx p b d jA jL jJ begin
        rc = ccall((:enlsip_gn_solve, LIB), Cint,
                   (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64,
                    Ptr{Float64}, Float64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Info},
                    Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
                   h.ptr, m, n, t, J, m, rx, At, max(n, 1), cx, ε_rank, -1, -1, p, b, d, info, jA, jL, jJ)
        check(h, rc)
    end
    i = info[]
    (i.status & 1) != 0 && throw(LinearAlgebra.SingularException(0))   # Julia's `\` would have thrown
    current_iter.rankA = i.rankA
    current_iter.rankJ2 = i.rankJ2
    current_iter.dimA = i.rankA
    current_iter.dimJ2 = i.rankJ2
    current_iter.b_gn = b
    current_iter.d_gn = d
    return p, DeviceQR(h, FACTOR_A, n), DeviceQR(h, FACTOR_L11, t), DeviceQR(h, FACTOR_J2, m)
end

"""
    d_gn_prefix(d_gn, k, n, rankA) -> view(d_gn, 1:k)

The leading `n - rankA` entries of `d_gn` agree with the reference's up to a sign per entry (the thin Q of `F_J2` is unique up
to column signs once the pivots are fixed), so every norm of a prefix `d_gn[1:k]` with `k <= n - rankA` is the reference's; the
entries beyond are an orthogonal transform of the reference's tail and only their norm as a whole is the same (SURVEY §7 H1).
The reference slices such prefixes in `search_direction_analys` (`d_gn[1:prev_dimJ2m1]`, src/enlsip_functions.jl:1230-1231),
`choose_subspace_dimensions` (:1166-1169) and `check_termination_criteria` (:2448-2452); with `dimJ2 <= n - t_prev` they never
reach past `n - rankA`.  Route those slices through this helper when the HIP backend is active: it asserts the bound instead
of silently mixing tail components into a norm.
"""
function d_gn_prefix(d_gn::AbstractVector{Float64}, k::Integer, n::Integer, rankA::Integer)
    @assert 0 <= k <= n - rankA "d_gn[1:$k] reaches past n - rankA = $(n - rankA): only the norm of the whole tail is defined there (src/enlsip_functions.jl:1230-1231)"
    return view(d_gn, 1:k)
end

"""
    sub_search_direction_hip(h, m, n, t, dimA, dimJ2, code) -> p, b, d

Re-entry of `sub_search_direction` on the resident factors with truncated dimensions
(src/enlsip_functions.jl:1253, subspace minimisation).  SURVEY §7 H1: callers that slice
`d_gn[1:k]` must keep `k <= n - rankA`: `d_gn_prefix` above asserts it.
"""
function sub_search_direction_hip(h::Handle, m::Integer, n::Integer, t::Integer, dimA::Integer, dimJ2::Integer, code::Integer)
    p = zeros(Float64, n); b = zeros(Float64, t); d = zeros(Float64, m)
    GC.@preserve p b d check(h, ccall((:enlsip_gn_resolve, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        h.ptr, 0, dimA, dimJ2, code, p, b, d))
    return p, b, d
end

"""
    first_lagrange_mult_estimate_hip!(h, λ, ∇fx, scaling_done, diag_scale, iter, ε_rank)

Device form of `first_lagrange_mult_estimate!` (src/enlsip_functions.jl:461-508) on the resident `F_A` and `cx`
of the last solve on `h`; writes `λ` and `iter.grad_res`.  Pass `∇fx = nothing` to use `Jᵀ rx` of the resident
`J`, `rx` (then `gradient_hip(h, n)` returns the same vector for the caller).
"""
function first_lagrange_mult_estimate_hip!(h::Handle, λ::Vector{Float64}, ∇fx::Union{Nothing,Vector{Float64}},
                                           scaling_done::Bool, diag_scale::Vector{Float64}, iter, ε_rank::Float64)
    gres = Ref{Float64}(0.0)
    g = ∇fx === nothing ? Ptr{Float64}(C_NULL) : pointer(∇fx)
    ds = scaling_done ? pointer(diag_scale) : Ptr{Float64}(C_NULL)
    GC.@preserve λ ∇fx diag_scale check(h, ccall((:enlsip_gn_first_lagrange, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ref{Float64}),
        h.ptr, 0, g, ds, ε_rank, λ, gres))
    iter.grad_res = gres[]
    return
end

"""
    second_lagrange_mult_estimate_hip!(h, λ, p_gn, scaling, diag_scale, ε_rank=sqrt(eps()))

Device form of `second_lagrange_mult_estimate!` (:514-537): uses the resident `J1 = (J*F_A.Q)[:, 1:t]` instead of
recomputing `J*F_A.Q` (:526).
"""
function second_lagrange_mult_estimate_hip!(h::Handle, λ::Vector{Float64}, p_gn::Vector{Float64}, scaling::Bool,
                                            diag_scale::Vector{Float64}, ε_rank::Float64=sqrt(eps(Float64)))
    ds = scaling ? pointer(diag_scale) : Ptr{Float64}(C_NULL)
    GC.@preserve λ p_gn diag_scale check(h, ccall((:enlsip_gn_second_lagrange, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
        h.ptr, 0, p_gn, ds, ε_rank, λ))
    return
end

"""    gradient_hip(h, n) -> Jᵀ rx of the J, rx of the last solve (src/enlsip_functions.jl:2690)"""
function gradient_hip(h::Handle, n::Integer)
    g = zeros(Float64, n)
    GC.@preserve g check(h, ccall((:enlsip_gn_gradient, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}), h.ptr, 0, g))
    return g
end

"""    factor_constraints_hip!(h, m, A, cx, ε_rank) -> (rankA, F_A, F_L11)

The constraint stage alone (src/enlsip_functions.jl:700, :768-769): leaves `F_A`, `F_L11` resident for
`first_lagrange_mult_estimate_hip!` (:704) before any subproblem solve.  `m` = rows of the solve that follows.
"""
function factor_constraints_hip!(h::Handle, m::Integer, A::Matrix{Float64}, cx::Vector{Float64}, ε_rank::Float64)
    t, n = size(A)
    At = Matrix(transpose(A))
    info = Ref(Info(0, 0, 0, 0, 0, 0))
    GC.@preserve At cx check(h, ccall((:enlsip_gn_factor_constraints, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Float64, Ref{Info}),
        h.ptr, m, n, t, At, max(n, 1), cx, ε_rank, info))
    return info[].rankA, DeviceQR(h, FACTOR_A, n), DeviceQR(h, FACTOR_L11, t)
end

"""    gn_search_direction_factored_hip!(h, J, rx, t, ε_rank, iter_k) -> (p_gn, F_A, F_L11, F_J2)

The solve of `update_working_set`'s `s == 0` branch (:768-771) right after `factor_constraints_hip!` with the same working set:
the resident `F_A`, `F_L11`, `b`, `p1` are reused, only `J` and `rx` are sent.
"""
function gn_search_direction_factored_hip!(h::Handle, J::Matrix{Float64}, rx::Vector{Float64}, t::Integer, ε_rank::Float64, iter_k)
    m, n = size(J)
    kA = min(n, t)
    p = zeros(Float64, n); b = zeros(Float64, max(t, 1)); d = zeros(Float64, m)
    jA = zeros(Int64, max(t, 1)); jL = zeros(Int64, max(kA, 1)); jJ = zeros(Int64, n)
    info = Ref(Info(0, 0, 0, 0, 0, 0))
    GC.@preserve J rx p b d jA jL jJ check(h, ccall((:enlsip_gn_solve_factored, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Float64, Int64, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ref{Info}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
        h.ptr, m, n, t, J, m, rx, ε_rank, -1, p, b, d, info, jA, jL, jJ))
    iter_k.rankA = info[].rankA; iter_k.rankJ2 = info[].rankJ2
    iter_k.dimA = info[].dimA; iter_k.dimJ2 = info[].dimJ2
    iter_k.b_gn = b[1:t]; iter_k.d_gn = d
    return p, DeviceQR(h, FACTOR_A, n), DeviceQR(h, FACTOR_L11, t), DeviceQR(h, FACTOR_J2, m)
end

"""    jacobian_times_hip(h, p, m, t) -> (J*p, C.A*p) on the J and A of the last solve (src/enlsip_functions.jl:2226-2229)"""
function jacobian_times_hip(h::Handle, p::Vector{Float64}, m::Integer, t::Integer)
    Jp = zeros(Float64, m); Ap = zeros(Float64, t)
    GC.@preserve p Jp Ap check(h, ccall((:enlsip_gn_jacobian_times, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h.ptr, 0, p, Jp, t > 0 ? pointer(Ap) : Ptr{Float64}(C_NULL)))
    return Jp, Ap
end

"""    full_constraints_times_hip(h, A, p) -> A * p

The product of the line-search set-up with the FULL constraint Jacobian, inactive rows included (`Ap = A * p`,
src/enlsip_functions.jl:2227), on the device: `A` (l x n) is staged, nothing about it has to be resident.
"""
function full_constraints_times_hip(h::Handle, A::Matrix{Float64}, p::Vector{Float64})
    l, n = size(A)
    Ap = zeros(Float64, l)
    l == 0 && return Ap
    GC.@preserve A p Ap check(h, ccall((:enlsip_gn_full_constraints_times, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}), h.ptr, l, n, A, l, p, Ap))
    return Ap
end

"""    route(h) -> Vector{String}: the kernel-selection branches the last solve on `h` took (enlsip_gn_get_route; diagnostics)"""
function route(h::Handle)
    mask = Ref{UInt64}(0)
    check(h, ccall((:enlsip_gn_get_route, LIB), Cint, (Ptr{Cvoid}, Ref{UInt64}), h.ptr, mask))
    names = String[]
    bit = 0
    while true
        nm = ccall((:enlsip_gn_route_name, LIB), Ptr{UInt8}, (Cint,), bit)
        nm == C_NULL && break
        (mask[] >> bit) & 1 == 1 && push!(names, unsafe_string(nm))
        bit += 1
    end
    return names
end

"""    tsqr_exchange(h) -> (transport, ranks, rank_tags_seen) of the last `gn_search_direction_tsqr_hip` on `h`:
`rank_tags_seen == ranks` on every rank means one correctly tagged message from each rank arrived (transport 1 = RCCL)."""
function tsqr_exchange(h::Handle)
    tr = Ref{Cint}(-1); rk = Ref{Cint}(0); seen = Ref{Cint}(-1)
    check(h, ccall((:enlsip_gn_tsqr_get_exchange, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}), h.ptr, tr, rk, seen))
    return Int(tr[]), Int(rk[]), Int(seen[])
end

"""    diagR(F::DeviceQR) -> diag(F.R) without moving the triangle (what `pseudo_rank(diag(F.R), ε)` needs, :768, :224)"""
function diagR(F::DeviceQR)
    r, c = factor_shape(F)
    k = min(r, c)
    dg = zeros(Float64, max(k, 1))
    GC.@preserve dg check(getfield(F, :h), ccall((:enlsip_gn_get_diagR, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}),
                                                 getfield(F, :h).ptr, getfield(F, :which), 0, dg))
    return dg[1:k]
end

"""    gn_search_direction_batched_hip(h, Js, rxs, As, cxs, ε_rank) -> (P, infos)

New capability (no reference counterpart): `B` independent subproblems of one shape in one call — `Js` is `m×n×B`, `rxs` `m×B`,
`As` `t×n×B` (the active constraint Jacobians), `cxs` `t×B`.  Returns the directions as the columns of `P` (`n×B`) and the
per-problem `Info` records.  Factors stay resident; pass `prob` to the accessors of the C ABI to reach problem `k`.
"""
function gn_search_direction_batched_hip(h::Handle, Js::Array{Float64,3}, rxs::Matrix{Float64}, As::Array{Float64,3},
                                         cxs::Matrix{Float64}, ε_rank::Float64)
    m, n, B = size(Js)
    t = size(As, 1)
    kA = min(n, t)
    Ats = permutedims(As, (2, 1, 3))                     # n×t×B: every slice is C.A' column-major
    P = zeros(Float64, n, B); b = zeros(Float64, max(t, 1), B); d = zeros(Float64, m, B)
    jA = zeros(Int64, max(t, 1), B); jL = zeros(Int64, max(kA, 1), B); jJ = zeros(Int64, n, B)
    infos = fill(Info(0, 0, 0, 0, 0, 0), B)
    GC.@preserve Js rxs Ats cxs P b d jA jL jJ infos check(h, ccall((:enlsip_gn_solve_batched, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Float64}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Int64,
         Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Info}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
        h.ptr, B, m, n, t, Js, m, m * n, rxs, Ats, max(n, 1), n * t, cxs, ε_rank, P, b, d, infos, jA, jL, jJ))
    return P, infos
end

"""    newton_search_direction_hip(h, Γ_mat) -> (p, error)

`newton_search_direction` (src/enlsip_functions.jl:348-423) after its two Hessian sums: the caller runs `hessian_res!` /
`hessian_cons!` (:391-394, callback-bound) and passes `Γ_mat = r_mat - c_mat`; the library does :398-421 on the resident
`F_A`, `p1`, `J` of the last solve.  Full-rank working sets only (`t == rankA`); the rank-deficient branch stays in Julia.
"""
function newton_search_direction_hip(h::Handle, Γ_mat::Matrix{Float64})
    n = size(Γ_mat, 1)
    p = zeros(Float64, n)
    bad = Ref{Int64}(0)
    GC.@preserve Γ_mat p check(h, ccall((:enlsip_gn_newton_direction, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ref{Int64}), h.ptr, 0, Γ_mat, n, p, bad))
    return p, bad[] != 0
end

# ---- one tall Jacobian, rows sharded over the GPUs of a node (config C4): the library's TSQR collective -------------------------
#
# One Julia process per GPU (Distributed / MPI.jl); every rank creates its Handle on its own device.  Rank 0 obtains the RCCL
# unique id, the caller's own transport carries its 128 bytes to the other ranks (e.g. `MPI.Bcast!(id, 0, comm)` or
# `remotecall_fetch`), then every rank calls `tsqr_init_rccl!` (collective: ncclCommInitRank inside the library) once.
# Afterwards `gn_search_direction_tsqr_hip` is `gn_search_direction` for a Jacobian whose rows live on several GPUs: rank g
# passes DEVICE pointers to its row block of J and rx (e.g. `pointer(::ROCArray)` from AMDGPU.jl) and the replicated C.A', cx;
# every rank gets the same p, ||d||, ranks and pivots back.  The one exchange step is an ncclAllGather of the packed triangles
# (8 n2 (n2 + 1) / 2 bytes per rank) over xGMI, issued by the library on the handle's stream.

"""    tsqr_unique_id() -> Vector{UInt8} (128 bytes; call on rank 0, broadcast with your own transport)"""
function tsqr_unique_id()
    id = zeros(UInt8, 128)
    rc = GC.@preserve id ccall((:enlsip_gn_tsqr_unique_id, LIB), Cint, (Ptr{UInt8},), id)
    rc == 0 || error("enlsip_gn_tsqr_unique_id failed with code $rc (RCCL not loadable?)")
    return id
end

"""    tsqr_init_rccl!(h, id, nranks, rank)  (collective over all ranks; rank is 0-based)"""
function tsqr_init_rccl!(h::Handle, id::Vector{UInt8}, nranks::Integer, rank::Integer)
    length(id) == 128 || error("the RCCL unique id has 128 bytes")
    GC.@preserve id check(h, ccall((:enlsip_gn_tsqr_init_rccl, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint),
                                   h.ptr, id, nranks, rank))
end

"""    gn_search_direction_tsqr_hip(h, dJ, ldj, drx, dAt, dcx, m_loc, n, t, ε_rank) -> (p, d_lead, d_norm, info, jpvtJ2)

Collective.  `dJ`, `drx`: device pointers to this rank's `m_loc` rows of `J` (column-major, leading dimension `ldj`) and `rx`;
`dAt`, `dcx`: device pointers to the replicated `C.A'` (`n×t`, column-major) and `C.cx` (C_NULL when `t == 0`).  Replaces
`JQ1 = J * F_A.Q` … `qr(J2, ColumnNorm())` … `sub_search_direction` (src/enlsip_functions.jl:219-225, :116-153) for that Jacobian;
`d_lead` = leading `n − rankA` entries of `F_J2.Q' d`, `d_norm` = `norm(d_gn)` over all ranks (what :1222-1231 and :2448 consume).
"""
function gn_search_direction_tsqr_hip(h::Handle, dJ::Ptr{Float64}, ldj::Integer, drx::Ptr{Float64}, dAt::Ptr{Float64},
                                      dcx::Ptr{Float64}, m_loc::Integer, n::Integer, t::Integer, ε_rank::Float64)
    p = zeros(Float64, n); dlead = zeros(Float64, n); jJ = zeros(Int64, n)
    dn = Ref{Float64}(0.0)
    info = Ref(Info(0, 0, 0, 0, 0, 0))
    GC.@preserve p dlead jJ check(h, ccall((:enlsip_gn_solve_tsqr, LIB), Cint,
        (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Float64,
         Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{Info}, Ptr{Int64}),
        h.ptr, m_loc, n, t, dJ, ldj, drx, dAt, max(n, 1), dcx, ε_rank, p, dlead, dn, info, jJ))
    (info[].status & 1) != 0 && throw(LinearAlgebra.SingularException(0))
    n2 = n - Int(info[].rankA)
    return p, dlead[1:n2], dn[], info[], jJ[1:n2]
end

end # module
