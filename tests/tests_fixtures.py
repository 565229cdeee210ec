"""Analytic fixtures the reference's own tests hold for the hot path (the formulas, restated)."""


def beta_bernoulli_posterior(obs_value: bool, alpha: float = 2.0, beta: float = 5.0):
    """p ~ Beta(alpha, beta), obs ~ Bernoulli(p)  =>  p | obs ~ Beta(alpha + obs, beta + 1 - obs)
    (reference tests/test_mcmc.py:118-137): -> (mean, variance, alpha', beta')."""
    a = alpha + float(obs_value)
    b = beta + (1.0 - float(obs_value))
    return a / (a + b), (a * b) / ((a + b) ** 2 * (a + b + 1)), a, b


def normal_normal_posterior(y_obs: float, prior_mean: float = 0.0, prior_var: float = 1.0, likelihood_var: float = 0.25):
    """mu ~ N(prior_mean, prior_var), y ~ N(mu, likelihood_var) (reference tests/test_mcmc.py:140-163):
    -> (posterior mean, posterior variance)."""
    prec = 1.0 / prior_var + 1.0 / likelihood_var
    return (prior_mean / prior_var + y_obs / likelihood_var) / prec, 1.0 / prec
