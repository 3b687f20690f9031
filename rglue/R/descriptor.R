## descriptor.R -- log-density DESCRIPTORS for the device path of metrop() and temper().
##
## An R closure cannot run on a GPU, so `obj` may also be a descriptor: a list of class "mcmc_descriptor" that names
## a log unnormalized density the device can evaluate.  metrop() and temper() are S3 generics (R/metrop.R:2-4,
## R/temper.R:2-4): the methods below are all that is needed, no existing R file changes.  Results keep the
## reference's names, order, shapes and classes; objects returned by the device path carry an extra class
## ("metropolis_gpu" / "tempering_gpu", in front of "metropolis" / "tempering") so that a restart
## metrop(out) / temper(out) dispatches here instead of to metrop.metropolis, which would call obj$lud as a function.
##
## NOT RUN IN THIS REPOSITORY'S IMAGE (no R); the logic is mirrored and tested in Python (mcmc_amd/api.py).

.descriptor <- function(kind, d, ...) {
    fields <- list(kind = kind, d = as.double(d), ...)
    for (nm in names(fields))
        if (is.numeric(fields[[nm]]) && nm != "models") storage.mode(fields[[nm]]) <- "double"
    fields$cache <- new.env(parent = emptyenv())   # holds the device-resident copy (an external pointer)
    structure(fields, class = "mcmc_descriptor")
}

## the vignette's posterior (vignettes/demo.Rnw:59-65): logistic regression, iid N(0, prior.sd^2) prior
logistic_descriptor <- function(x, y, prior.sd = 2) {
    x <- as.matrix(x)
    stopifnot(is.numeric(x), is.numeric(y), length(y) == nrow(x), all(y %in% c(0, 1)))
    .descriptor("logistic", ncol(x), x = x, y = as.double(y), prior.sd = prior.sd)
}

## -(x - mu)' P (x - mu) / 2
mvnorm_descriptor <- function(mu, precision) {
    precision <- as.matrix(precision)
    stopifnot(nrow(precision) == length(mu), ncol(precision) == length(mu), isSymmetric(precision))
    .descriptor("mvnorm", length(mu), mu = mu, precision = precision)
}

## finite mixture of isotropic normals: weights (K), means (K x d), sds (K or 1)
normmix_descriptor <- function(weights, means, sds) {
    means <- as.matrix(means)
    K <- nrow(means)
    stopifnot(length(weights) == K, all(weights > 0), all(sds > 0), length(sds) %in% c(1, K))
    .descriptor("normmix", ncol(means), weights = weights, means = means, sds = rep_len(sds, K))
}

## heavy-tailed test target: sum(dt(x, df, log = TRUE))  (tests/morph.metrop.R:9)
student_descriptor <- function(d, df) .descriptor("student.t", d, df = df)

## tempering targets.  witch's hat of tests/temp-par-witch.R:24-40; tempered mixture h(i, x) = beta_i * normmix(x);
## the model-selection posterior of tests/temp-par.R:39-56 / vignettes/bfst.Rnw:472-489
witch_descriptor <- function(d, bounds) .descriptor("witch", d, bounds = bounds)
tempered_normmix_descriptor <- function(betas, weights, means, sds) {
    means <- as.matrix(means)
    .descriptor("tempered.normmix", ncol(means), betas = betas, weights = weights, means = means,
        sds = rep_len(sds, nrow(means)))
}
logistic_models_descriptor <- function(x, y, models, prior.sd = 1, log.pseudo.prior = NULL, constant = 0) {
    x <- as.matrix(x)
    models <- cbind(1L, matrix(as.integer(models), nrow(as.matrix(models))))   # column 1: the intercept
    stopifnot(ncol(models) == ncol(x), length(y) == nrow(x))
    if (is.null(log.pseudo.prior)) log.pseudo.prior <- double(nrow(models))
    stopifnot(length(log.pseudo.prior) == nrow(models))
    .descriptor("logistic.models", ncol(x), x = x, y = as.double(y), prior.sd = prior.sd, models = models,
        log.pseudo.prior = log.pseudo.prior, constant = constant)
}

## the Philox key of the device stream comes from R's generator, so set.seed() governs the device path too
.new_device_seed <- function() c(floor(runif(2) * 2^32), 0, 0)

metrop.mcmc_descriptor <- function(obj, initial, nbatch, blen = 1, nspac = 1, scale = 1, outfun, debug = FALSE,
    device.seed = NULL, ...)
{
    if (! exists(".Random.seed")) runif(1)
    saveseed <- .Random.seed
    if (is.null(device.seed)) device.seed <- .new_device_seed()
    func2 <- if (missing(outfun)) NULL else outfun
    if (missing(outfun)) outfun <- NULL
    out.time <- system.time(
    out <- .Call(C_metrop_descriptor, obj, initial, nbatch, blen, nspac, scale, func2, debug,
        as.double(device.seed))
    )
    out$initial.seed <- saveseed
    out$final.seed <- .Random.seed
    out$time <- out.time
    out$lud <- obj
    out$nbatch <- nbatch
    out$blen <- blen
    out$nspac <- nspac
    out$scale <- scale
    out$outfun <- outfun
    out$debug <- debug
    swap12 <- function(a) if (length(dim(a)) == 3) aperm(a, c(2, 1, 3)) else t(a)
    out$batch <- swap12(out$batch)
    for (nm in c("current", "proposal", "z"))
        if (! is.null(out[[nm]])) out[[nm]] <- swap12(out[[nm]])
    class(out) <- c("mcmc", "metropolis_gpu", "metropolis")
    out
}

## restart: missing arguments come from the object, the run starts at obj$final and continues the same device
## stream (same key, iteration counter advanced), so that split runs equal one long run (tests/saveseed.R:12-17)
metrop.metropolis_gpu <- function(obj, initial, nbatch, blen = 1, nspac = 1, scale = 1, outfun, debug = FALSE, ...)
{
    if (missing(nbatch)) nbatch <- obj$nbatch
    if (missing(blen)) blen <- obj$blen
    if (missing(nspac)) nspac <- obj$nspac
    if (missing(scale)) scale <- obj$scale
    if (missing(debug)) debug <- obj$debug
    assign(".Random.seed", obj$final.seed, .GlobalEnv)
    if (missing(outfun)) outfun <- obj$outfun
    if (is.null(outfun))
        metrop.mcmc_descriptor(obj$lud, obj$final, nbatch, blen, nspac, scale, debug = debug,
            device.seed = obj$device.seed, ...)
    else
        metrop.mcmc_descriptor(obj$lud, obj$final, nbatch, blen, nspac, scale, outfun, debug,
            device.seed = obj$device.seed, ...)
}

temper.mcmc_descriptor <- function(obj, initial, neighbors, nbatch, blen = 1, nspac = 1, scale = 1, outfun,
    debug = FALSE, parallel = FALSE, device.seed = NULL, ...)
{
    if (! exists(".Random.seed")) runif(1)
    saveseed <- .Random.seed
    if (is.null(device.seed)) device.seed <- .new_device_seed()
    func2 <- if (missing(outfun)) NULL else outfun
    if (missing(outfun)) outfun <- NULL
    stopifnot(is.numeric(initial))
    storage.mode(initial) <- "double"
    if (is.list(scale)) {
        for (i in seq_along(scale)) {
            stopifnot(is.numeric(scale[[i]]))
            storage.mode(scale[[i]]) <- "double"
        }
    } else {
        stopifnot(is.numeric(scale))
        storage.mode(scale) <- "double"
    }
    out.time <- system.time(
    out <- .Call(C_temper_descriptor, obj, initial, neighbors, nbatch, blen, nspac, scale, func2, debug, parallel,
        as.double(device.seed))
    )
    structure(c(list(lud = obj, neighbors = neighbors, nbatch = nbatch, blen = blen, nspac = nspac, scale = scale,
        outfun = outfun, debug = debug, parallel = parallel, initial.seed = saveseed, final.seed = .Random.seed,
        time = out.time), out), class = c("mcmc", "tempering_gpu", "tempering"))
}

temper.tempering_gpu <- function(obj, initial, neighbors, nbatch, blen = 1, nspac = 1, scale = 1, outfun,
    debug = FALSE, parallel = FALSE, ...)
{
    if (missing(nbatch)) nbatch <- obj$nbatch
    if (missing(blen)) blen <- obj$blen
    if (missing(nspac)) nspac <- obj$nspac
    if (missing(scale)) scale <- obj$scale
    if (missing(debug)) debug <- obj$debug
    assign(".Random.seed", obj$final.seed, .GlobalEnv)
    if (missing(outfun)) outfun <- obj$outfun
    if (is.null(outfun))
        temper.mcmc_descriptor(obj$lud, obj$final, obj$neighbors, nbatch, blen, nspac, scale, debug = debug,
            parallel = obj$parallel, device.seed = obj$device.seed, ...)
    else
        temper.mcmc_descriptor(obj$lud, obj$final, obj$neighbors, nbatch, blen, nspac, scale, outfun, debug = debug,
            parallel = obj$parallel, device.seed = obj$device.seed, ...)
}

## the pseudo-prior adjustment loop of vignettes/bfst.Rnw:497-559 for serial tempering on the device: run, raise the
## log pseudo-prior of the components visited less often, qux <- qux + pmin(log(max(ibar) / ibar), 10) (516), shift it
## to minimum zero (517), continue from the last state (544) and stop once max(ibar) / min(ibar) < 2 (550)
tune_pseudo_prior <- function(desc, initial, neighbors, nbatch, blen = 1, nspac = 1, scale = 1,
    max.rounds = 50, ratio = 2, cap = 10, ...)
{
    stopifnot(inherits(desc, "mcmc_descriptor"), desc$kind == "logistic.models")
    qux <- desc$log.pseudo.prior
    qux.save <- NULL
    ibar.save <- NULL
    device.seed <- NULL
    with_pseudo_prior <- function(desc, qux) {
        fields <- unclass(desc)
        fields$cache <- new.env(parent = emptyenv())    # a new device copy: the parameter block changed
        fields$log.pseudo.prior <- as.double(qux)
        structure(fields, class = "mcmc_descriptor")
    }
    for (round in seq_len(max.rounds)) {
        desc <- with_pseudo_prior(desc, qux)
        out <- temper(desc, initial = initial, neighbors = neighbors, nbatch = nbatch, blen = blen, nspac = nspac,
            scale = scale, parallel = FALSE, device.seed = device.seed, ...)
        device.seed <- out$device.seed
        initial <- out$final
        ibar <- colMeans(out$ibatch)
        ibar.save <- rbind(ibar.save, ibar, deparse.level = 0)
        qux.run <- qux
        qux <- qux + pmin(log(max(ibar) / ibar), cap)
        qux <- qux - min(qux)
        qux.save <- rbind(qux.save, qux, deparse.level = 0)
        if (max(ibar) / min(ibar) < ratio) break      # after the update, as vignettes/bfst.Rnw:546-550
    }
    list(log.pseudo.prior = qux, log.pseudo.prior.run = qux.run, qux.save = qux.save, ibar.save = ibar.save, out = out,
        descriptor = with_pseudo_prior(desc, qux))
}
